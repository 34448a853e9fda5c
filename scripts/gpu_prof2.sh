#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c_biggan -o prof -- python $R/scripts/run_leg.py biggan128 3 > $R/gpurun_out/prof_c_biggan.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c_dstep -o prof -- python $R/scripts/run_leg.py resnet128_dstep 5 > $R/gpurun_out/prof_c_dstep.log 2>&1
cd $R
find gpurun_out/prof_c_biggan gpurun_out/prof_c_dstep -name "*.db" -delete 2>/dev/null
find gpurun_out/prof_c_biggan gpurun_out/prof_c_dstep -name "*kernel_trace.csv" -delete 2>/dev/null
tail -2 gpurun_out/prof_c_biggan.log | cut -c1-300
