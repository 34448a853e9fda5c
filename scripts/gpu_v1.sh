#!/bin/bash
# Round-2 evidence visit: rocprofv3 kernel stats of the default bench workload (cifar step) and of
# the resnet128 D-step leg, the two PMC passes for HBM traffic, per-geometry conv timings.
# usage: gpu_v1.sh TAG
TAG=${1:-v1}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 200 python scripts/prof_leg_shapes.py resnet128_dstep > gpurun_out/${TAG}_shapes_dstep.txt 2>&1
timeout 200 python scripts/prof_leg_shapes.py cifar > gpurun_out/${TAG}_shapes_cifar.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_cifar -o prof -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-fid --no-legs > $R/gpurun_out/prof_${TAG}_cifar.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_dstep -o prof -- python $R/scripts/run_leg.py resnet128_dstep 5 > $R/gpurun_out/prof_${TAG}_dstep.log 2>&1
rm -rf /tmp/pmc_f /tmp/pmc_w
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o p -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-fid --no-legs > $R/gpurun_out/pmc_fetch_$TAG.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o p -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-fid --no-legs > $R/gpurun_out/pmc_write_$TAG.log 2>&1
cd $R
python scripts/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w gpurun_out/pmc_traffic_$TAG.json > gpurun_out/pmc_traffic_$TAG.txt 2>&1
find gpurun_out/prof_${TAG}_cifar gpurun_out/prof_${TAG}_dstep -name "*.db" -delete 2>/dev/null
find gpurun_out/prof_${TAG}_cifar gpurun_out/prof_${TAG}_dstep -name "*kernel_trace.csv" -delete 2>/dev/null
tail -1 gpurun_out/prof_${TAG}_cifar.log | cut -c1-300
tail -4 gpurun_out/${TAG}_shapes_dstep.txt
cat gpurun_out/pmc_traffic_$TAG.txt | tail -20
