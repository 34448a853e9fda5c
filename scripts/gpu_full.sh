#!/bin/bash
# full GPU test-suite + bench legs.  usage: gpu_full.sh TAG [bench args]
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/full_${TAG}_tests.txt
cat gpurun_out/full_${TAG}_tests.txt
timeout 900 python bench.py "$@" 2>gpurun_out/full_${TAG}_bench.err | tail -1 > gpurun_out/full_${TAG}_bench.json
python - <<PY
import json
d=json.load(open('gpurun_out/full_${TAG}_bench.json'))
print('cifar', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))
for leg in ['resnet128_dstep','resnet128_dstep_gp','biggan128']:
    L=d.get(leg)
    if not L: continue
    if 'error' in L: print(leg, L); continue
    print(leg, L['ms'], L['tflops'], L['frac'], 'conv ms', L['conv_kernel_ms_eager'])
    for k,v in L['kernels'].items(): print('    %-34s %8.3f ms %7.1f us %7.1f TF  x%d' % (k, v['ms_per_step'], v['avg_launch_us'], v['tflops'], v['launches_per_step']))
PY
