#!/bin/bash
# same-box A/B of two builds of the library over a list of convolution shapes:
#   bash scripts/ab_convs.sh TAG "SHAPES" [reps]     (prev = lib/libcgamd_prev.so, cur = lib/libcgamd.so)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R" || exit 1; mkdir -p gpurun_out
TAG=$1; SH=$2; REPS=${3:-2}
: > gpurun_out/${TAG}_ab.txt
for rep in $(seq 1 $REPS); do
  for which in prev cur; do
    LIBP=$R/compare_gan_amd/lib/libcgamd.so; [ $which = prev ] && LIBP=$R/compare_gan_amd/lib/libcgamd_prev.so
    echo "## $which (rep $rep)" >> gpurun_out/${TAG}_ab.txt
    CGAMD_LIB_PATH=$LIBP timeout 600 python scripts/bench_convs.py "$SH" 2>&1 | grep -v amdgpu.ids >> gpurun_out/${TAG}_ab.txt
  done
done
cat gpurun_out/${TAG}_ab.txt
