#!/bin/bash
# kernel-stats profile of the cifar step + the D-step leg.  usage: gpu_prof.sh TAG
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 300 python scripts/run_leg.py resnet128_dstep 10 2>/dev/null | tail -1 > gpurun_out/prof_${TAG}_dstep.json
python - <<PY
import json
L=json.load(open('gpurun_out/prof_${TAG}_dstep.json'))
print('dstep', L['ms'], L['tflops'], L['frac'], 'conv ms', L['conv_kernel_ms_eager'])
for k,v in L['kernels'].items(): print('    %-34s %8.3f ms %7.1f us %7.1f TF  x%d' % (k, v['ms_per_step'], v['avg_launch_us'], v['tflops'], v['launches_per_step']))
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_cifar -o prof -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-fid --no-legs > $R/gpurun_out/prof_${TAG}_cifar.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_dstep -o prof -- python $R/scripts/run_leg.py resnet128_dstep 5 > $R/gpurun_out/prof_${TAG}_dstep.log 2>&1
cd $R
find gpurun_out/prof_${TAG}_cifar gpurun_out/prof_${TAG}_dstep -name "*.db" -delete 2>/dev/null
find gpurun_out/prof_${TAG}_cifar gpurun_out/prof_${TAG}_dstep -name "*kernel_trace.csv" -delete 2>/dev/null
tail -1 gpurun_out/prof_${TAG}_cifar.log | cut -c1-300
