#!/bin/bash
# Profiles of bench.py on one MI355X: kernel-trace statistics, then the counter passes, each in a run of its own.
#   tools/profile_bench.sh <tag> [bench.py arguments]     e.g.  tools/profile_bench.sh r02_c3 --workload c3
# Results land under gpurun_out/<tag>_*; the summaries meant to be kept are copied into profiles/ by hand.
set -u
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
run() { echo "== $*" >> "$out/${tag}_profile.log"; "$@" >> "$out/${tag}_profile.log" 2>&1; }
run timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_trace" -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-e2e "$@" || exit 1
for pass in "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
            "wait:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
            "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  run timeout -s KILL 200 rocprofv3 --pmc $ctrs --output-format csv -d "$out/${tag}_pmc_$name" -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-e2e "$@" || exit 1
  python3 tools/pmc_agg.py "$out/${tag}_pmc_$name" > "$out/${tag}_pmc_$name.json"
done
find "$out/${tag}_trace" -name '*kernel_stats.csv' -exec cp {} "$out/${tag}_kernel_stats.csv" \;
echo done >> "$out/${tag}_profile.log"
