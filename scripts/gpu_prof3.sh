#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for v in 0 1; do
CGAMD_FUSED_POOL=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_d_pool$v -o prof -- python $R/scripts/run_leg.py resnet128_dstep 5 > $R/gpurun_out/prof_d_pool$v.log 2>&1
done
cd $R
find gpurun_out/prof_d_pool0 gpurun_out/prof_d_pool1 -name "*.db" -delete 2>/dev/null
find gpurun_out/prof_d_pool0 gpurun_out/prof_d_pool1 -name "*kernel_trace.csv" -delete 2>/dev/null
