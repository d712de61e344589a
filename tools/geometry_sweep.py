#!/usr/bin/env python3
"""Times the painting of K = N targets for several N and every wave count per target (bench.py children).

Writes one line per run to the file given as first argument: N, waves asked (0 = automatic), ms per step.
Stops at the first child that fails.  The parent never touches the GPU."""
import json
import subprocess
import sys

out = sys.argv[1]
sizes = [int(x) for x in sys.argv[2].split(",")]
waves = [int(x) for x in sys.argv[3].split(",")]
extra = sys.argv[4:]
with open(out, "a") as f:
    for n in sizes:
        for w in waves:
            if w and (n + 63) // 64 > 80 * w:
                continue
            cmd = ["timeout", "-k", "10", "120", sys.executable, "bench.py", "--no-cpu", "--steps", "3", "--warmup", "1",
                   "--N", str(n), "--waves", str(w)] + extra
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                f.write(f"{n} {w} FAILED rc={r.returncode} {r.stderr[-300:]!r}\n")
                sys.exit(1)
            line = json.loads(r.stdout.strip().splitlines()[-1])
            f.write(f"{n} {w} {line['ms_per_step']:.3f} {line['config']}\n")
            f.flush()
            print(n, w, line["ms_per_step"], flush=True)
