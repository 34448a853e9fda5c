#!/bin/bash
# Quick GPU visit: op tests, graph bench (no cpu baseline), rocprof kernel stats.  Usage: gpu_quick.sh TAG [pytest -k expr]
TAG=${1:-q}
KEXPR=${2:-""}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
if [ -n "$KEXPR" ]; then
  timeout 600 python -m pytest tests -m gpu -q -x -k "$KEXPR" 2>&1 | tail -15 > gpurun_out/tests_$TAG.log
else
  timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/tests_$TAG.log
fi
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_$TAG.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-fid > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
rm -f gpurun_out/prof_$TAG/*kernel_trace.csv
tail -4 gpurun_out/tests_$TAG.log; cut -c1-900 gpurun_out/bench_$TAG.log
