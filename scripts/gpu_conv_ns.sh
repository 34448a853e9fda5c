#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
run() { echo "== $1 $2"; env $1 timeout 100 python scripts/bench_convs.py $2 2>&1 | grep -v amdgpu | cut -c1-70; }
{
run CGAMD_CONV_NS=0 fixed
run CGAMD_CONV_NS=1 fixed
run CGAMD_CONV_NS=0 resnet128
run CGAMD_CONV_NS=1 resnet128
} > gpurun_out/conv_ns.txt 2>&1
cat gpurun_out/conv_ns.txt
CGAMD_CONV_NS=1 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fid --no-roofline 2>&1 | tail -1 | cut -c1-200
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fid --no-roofline 2>&1 | tail -1 | cut -c1-200
