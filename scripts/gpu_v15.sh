#!/bin/bash
# last visit of round 2: smoke(), the bench line of the final code
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/final3_bench.json 2> gpurun_out/final3_bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/final3_bench.json').read().strip().splitlines()[-1])
print('cifar', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'])
print('fid10k', d['fid10k']['wall_s'], d['fid10k'].get('extractor_setup_s'), d['fid10k']['split_s'])
print('cpu', d['cpu_baseline']['value'])
for leg in ['resnet128_dstep','resnet128_dstep_gp','biggan128']:
    L=d.get(leg)
    if L: print(leg, L.get('ms'), L.get('tflops'), L.get('frac'), L.get('error'))
PY
