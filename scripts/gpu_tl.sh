#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export CGAMD_LIB_PATH=$R/compare_gan_amd/lib/libcgamd_timing.so
for s in "$@"; do
  timeout 120 python scripts/hconv_timeline.py $s >> gpurun_out/hconv_timeline.txt 2>&1
done
cat gpurun_out/hconv_timeline.txt
