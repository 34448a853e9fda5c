#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats.  Usage: scripts/gpu_round.sh TAG
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/tests_$TAG.log 2>&1
echo "pytest rc=$?" >> gpurun_out/tests_$TAG.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_$TAG.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench_$TAG.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof rc=$?" >> $R/gpurun_out/prof_$TAG.log
cd $R
find gpurun_out/prof_$TAG -name "*.db" -delete 2>/dev/null
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
tail -5 gpurun_out/tests_$TAG.log; tail -3 gpurun_out/bench_$TAG.log; tail -3 gpurun_out/prof_$TAG.log
