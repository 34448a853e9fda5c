#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for m in 200 100 32; do
  echo "== CGAMD_HCONV_MIN=$m CGAMD_HWGRAD_MIN=$m" 
  CGAMD_HCONV_MIN=$m CGAMD_HWGRAD_MIN=$m timeout 300 python scripts/bench_convs.py cifar 2>&1 | grep -v amdgpu.ids
done > gpurun_out/thr_cifar.txt
cat gpurun_out/thr_cifar.txt
