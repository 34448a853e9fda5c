#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
run() { echo "== $1"; env $1 timeout 100 python scripts/bench_convs.py fixed 2>&1 | grep -v amdgpu | cut -c1-70; }
{
run CGAMD_CONV_DBG=0
run CGAMD_CONV_DBG=1
run CGAMD_CONV_DBG=2
run CGAMD_CONV_DBG=4
run CGAMD_CONV_DBG=6
run "CGAMD_CONV_NS=2"
run "CGAMD_CONV_NS=4"
} > gpurun_out/conv_dbg.txt 2>&1
cat gpurun_out/conv_dbg.txt
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fid 2>&1 | tail -1 | cut -c1-200
