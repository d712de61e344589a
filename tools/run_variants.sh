#!/bin/bash
# On the GPU box: for every variant library under relate_amd/csrc/exp/ named on the command line, the parity tests
# that reach its geometries and one c3 bench line (kernel only) -> gpurun_out/variants/NAME.{log,json}.
# Stops at the first failure (nothing further is launched after a fault or a time limit).
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/variants
export TAG
for v in "$@"; do
  lib=$PWD/relate_amd/csrc/exp/$v/librelate_gpu.so
  [ -f "$lib" ] || { echo "no $lib"; exit 1; }
  echo "== $v: tests"
  RELATE_GPU_LIB=$lib timeout -k 10 400 python -u -m pytest tests/test_gpu_geometries.py tests/test_gpu_fuzz.py -m gpu -x -q \
    --timeout 120 --timeout-method thread > gpurun_out/variants/$v.log 2>&1 || { tail -n 30 gpurun_out/variants/$v.log; exit 1; }
  tail -n 2 gpurun_out/variants/$v.log
  echo "== $v: bench"
  RELATE_GPU_LIB=$lib timeout -k 10 300 python -u bench.py --no-cpu --no-e2e --steps 5 --warmup 2 ${BENCH_ARGS:-} > gpurun_out/variants/$v$TAG.json 2>> gpurun_out/variants/$v.log \
    || { tail -n 30 gpurun_out/variants/$v.log; exit 1; }
  python - "$v" <<'PY'
import json, sys
import os
tag = os.environ.get("TAG", "")
d = json.loads(open(f"gpurun_out/variants/{sys.argv[1]}{tag}.json").read().strip().splitlines()[-1])
print(sys.argv[1] + tag, "ms_per_step", round(d["ms_per_step"], 3), "kernel_ms", round(d["roofline"]["kernel_ms"], 3), d["config"].get("kernel_geometry"))
PY
done
