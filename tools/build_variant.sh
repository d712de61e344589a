#!/bin/bash
# Builds librelate_gpu.so with extra -D switches for the register-resident kernel files (experiments):
#   tools/build_variant.sh NAME -DRELATE_EXP_PRIO=1 ...   ->  relate_amd/csrc/exp/NAME/{librelate_gpu.so, paint_regs_g*.s}
# GS="4" limits the rebuilt kernel files (default: 2 4 8, the geometries the switches act on).
# The other objects are taken from the regular build (run `make -C relate_amd/csrc` first).  Use with
# RELATE_GPU_LIB=relate_amd/csrc/exp/NAME/librelate_gpu.so.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/relate_amd/csrc
out=$src/exp/$name
mkdir -p "$out"
flags="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-result"
pids=()
for g in ${GS:-2 4 8}; do
  (
    tmp=$out/.tmp_g$g; rm -rf "$tmp"; mkdir -p "$tmp"
    hipcc $flags "$@" -I"$src" -save-temps=obj -c "$src/paint_regs_g$g.hip" -o "$tmp/paint_regs_g$g.o"
    mv "$tmp/paint_regs_g$g-hip-amdgcn-amd-amdhsa-gfx950.s" "$out/paint_regs_g$g.s"
    python3 "$root/tools/check_regs.py" "$out/paint_regs_g$g.s"
    mv "$tmp/paint_regs_g$g.o" "$out/paint_regs_g$g.o"
    rm -rf "$tmp"
  ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
objs=()
for g in 1 2 4 8; do
  if [ -f "$out/paint_regs_g$g.o" ]; then objs+=("$out/paint_regs_g$g.o"); else objs+=("$src/paint_regs_g$g.o"); fi
done
hipcc --offload-arch=gfx950 -shared -o "$out/librelate_gpu.so" "$src/paint_kernels.o" "${objs[@]}" \
  "$src/relate_gpu.o" "$src/chunk_io.o" "$src/synth.o" "$src/haps_loader.o"
echo "$out/librelate_gpu.so"
