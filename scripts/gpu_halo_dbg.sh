#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
run() { echo "== $1"; env $1 timeout 100 python scripts/bench_convs.py halo 2>&1 | grep -v amdgpu | awk '{print $1, $NF-0, $(NF-1)}'; }
{
run CGAMD_HALO_BLOCKS=512
run CGAMD_HALO_BLOCKS=256
run CGAMD_HALO_BLOCKS=1024


run CGAMD_NO_HALO_WGRAD=1
} > gpurun_out/halo_dbg.txt 2>&1
cat gpurun_out/halo_dbg.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gconv or conv or stem" 2>&1 | tail -3
