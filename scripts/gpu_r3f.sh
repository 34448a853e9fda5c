#!/bin/bash
# round 3: kernel stats of the cifar step and the D-step leg with the small-map kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_cifar -o prof -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-fid --no-legs > $R/gpurun_out/r3f_cifar.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dstep -o prof -- python $R/scripts/run_leg.py resnet128_dstep 5 > $R/gpurun_out/r3f_dstep.log 2>&1
cd $R
cp $(find /tmp/p_cifar -name "*kernel_stats.csv" | head -1) gpurun_out/r3f_cifar_kernel_stats.csv
cp $(find /tmp/p_dstep -name "*kernel_stats.csv" | head -1) gpurun_out/r3f_dstep_kernel_stats.csv
tail -2 gpurun_out/r3f_cifar.log | cut -c1-300
