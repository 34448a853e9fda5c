#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fid --no-roofline"
{
for v in 256 257 513 1025; do echo "T128_MIN=$v"; CGAMD_CONV_T128_MIN=$v timeout 200 $B 2>&1 | tail -1 | cut -c1-190; done
} > gpurun_out/t128.txt 2>&1
cat gpurun_out/t128.txt
timeout 300 python -m pytest tests -m gpu -q -x -k "modular or train or wgangp or biggan or loss" 2>&1 | tail -3
