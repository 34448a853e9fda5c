#!/bin/bash
# round 3: kernel stats of the BigGAN-128 step (eager, 3 steps)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_big -o prof -- python $R/scripts/run_leg_eager.py biggan128 3 > $R/gpurun_out/r3p_biggan.log 2>&1
cd $R
cp $(find /tmp/p_big -name "*kernel_stats.csv" | head -1) gpurun_out/r3p_biggan_kernel_stats.csv
tail -2 gpurun_out/r3p_biggan.log | cut -c1-300
head -40 gpurun_out/r3p_biggan_kernel_stats.csv | cut -c1-160
