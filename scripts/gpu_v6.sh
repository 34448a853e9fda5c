#!/bin/bash
# Visit v6: failing tests again with full output, grey-image conv cases, block Jacobi tests + FID timing,
# rocprof kernel stats of the FID-10k leg
TAG=${1:-v6}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "grey or syevj or fid" 2>&1 > gpurun_out/${TAG}_tests_k.txt
tail -5 gpurun_out/${TAG}_tests_k.txt; grep -n "^E  " gpurun_out/${TAG}_tests_k.txt | head -20
timeout 900 python -m pytest tests/test_architectures_gpu.py tests/test_ssgan_gpu.py tests/test_eval_gpu.py -m gpu -q -s -k "architecture or resnet_stl or ssgan or fid or inception_features" > gpurun_out/${TAG}_tests.txt 2>&1
grep -n "max / mean\|worst\|ssgan\|passed\|failed\|^E  \|Error" gpurun_out/${TAG}_tests.txt | head -60
for v in 0 256; do
  echo "== CGAMD_JACOBI_BLOCK_MIN=$v"
  CGAMD_JACOBI_BLOCK_MIN=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('fid10k', d['fid10k']['wall_s'], d['fid10k']['split_s'], d['fid10k']['fid'])"
done 2>&1 | tee gpurun_out/${TAG}_jacobi_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_fid -o prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-legs > $R/gpurun_out/prof_${TAG}_fid.log 2>&1
cd $R
find gpurun_out/prof_${TAG}_fid -name "*.db" -delete 2>/dev/null; find gpurun_out/prof_${TAG}_fid -name "*kernel_trace.csv" -delete 2>/dev/null
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_${TAG}_fid/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:28]:
    print('%-90s %7s %9.1f us %9.2f ms' % (r['Name'][:90].replace('(anonymous namespace)::',''), r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
