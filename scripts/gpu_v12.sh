#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/final2_bench.json 2> gpurun_out/final2_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final2_fid -o prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-legs > $R/gpurun_out/prof_final2_fid.log 2>&1
cd $R
find gpurun_out/prof_final2_fid -name "*.db" -delete 2>/dev/null; find gpurun_out/prof_final2_fid -name "*kernel_trace.csv" -delete 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/final2_bench.json').read().strip().splitlines()[-1])
print('cifar', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'])
print('fid10k', d['fid10k']['wall_s'], d['fid10k'].get('extractor_setup_s'), d['fid10k']['split_s'])
print('cpu', d['cpu_baseline']['value'])
for leg in ['resnet128_dstep','resnet128_dstep_gp','biggan128']:
    L=d.get(leg)
    if L: print(leg, L.get('ms'), L.get('tflops'), L.get('frac'), L.get('error'))
PY
