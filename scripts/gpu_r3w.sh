#!/bin/bash
# round 3: kernel stats of the FID-10k leg
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_fid -o prof -- python $R/bench.py --steps 2 --warmup 1 --preheat-s 0 --no-cpu-baseline --no-roofline --no-legs > $R/gpurun_out/r3w_fid.log 2>&1
cd $R
cp $(find /tmp/p_fid -name "*kernel_stats.csv" | head -1) gpurun_out/r3w_fid_kernel_stats.csv
tail -1 gpurun_out/r3w_fid.log | cut -c1-200
head -25 gpurun_out/r3w_fid_kernel_stats.csv | cut -c1-150
