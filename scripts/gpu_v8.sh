#!/bin/bash
# Visit v8: targeted tests after the fixes, Jacobi diagnostics, a full bench line, kernel stats
TAG=${1:-v8}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_architectures_gpu.py tests/test_ssgan_gpu.py tests/test_eval_gpu.py tests/test_kernels_gpu.py -m gpu -q -s -k "architecture or resnet_stl or ssgan or fid or syevj or fp64 or kid or evaluate or grey" > gpurun_out/${TAG}_tests.txt 2>&1
grep -n "max / mean\|worst\|passed\|failed\|^E  \|Error" gpurun_out/${TAG}_tests.txt | head -40
timeout 300 python scripts/debug_jacobi.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_jacobi_debug.txt
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1])
print('cifar', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'])
print('fid10k', d['fid10k']['wall_s'], d['fid10k']['split_s'])
print('cpu', d['cpu_baseline'])
for leg in ['resnet128_dstep','resnet128_dstep_gp','biggan128']:
    L=d.get(leg)
    if not L: continue
    if 'error' in L: print(leg, L); continue
    print(leg, L['ms'], L['tflops'], L['frac'])
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_cifar -o prof -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-fid --no-legs > $R/gpurun_out/prof_${TAG}_cifar.log 2>&1
cd $R
find gpurun_out/prof_${TAG}_cifar -name "*.db" -delete 2>/dev/null; find gpurun_out/prof_${TAG}_cifar -name "*kernel_trace.csv" -delete 2>/dev/null
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_${TAG}_cifar/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('total kernel ms', tot/1e6, 'calls', calls)
for r in rows[:30]:
    print('%-90s %7s %9.1f us %9.2f ms' % (r['Name'][:90].replace('(anonymous namespace)::',''), r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
