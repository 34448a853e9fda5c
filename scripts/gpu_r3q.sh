#!/bin/bash
# round 3: per-launch geometry log (CGAMD_PROF_LOG) of the D-step leg's convolutions
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; rm -f gpurun_out/r3q_dstep_launches.txt
CGAMD_PROF_LOG=$R/gpurun_out/r3q_dstep_launches.txt timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fid --no-roofline --legs resnet128_dstep > gpurun_out/r3q_bench.json 2> gpurun_out/r3q_bench.err
grep "U2" gpurun_out/r3q_dstep_launches.txt | sort | uniq -c | sort -k2 | head -30
