#!/bin/bash
# Round-2 closing visit: full parity suite, smoke(), the bench line with all legs, forced-DP bench,
# rocprofv3 kernel stats (cifar step, D-step leg, FID leg), the two PMC passes for HBM traffic.
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/${TAG}_tests.txt 2>&1
tail -22 gpurun_out/${TAG}_tests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > gpurun_out/${TAG}_smoke.txt; cat gpurun_out/${TAG}_smoke.txt
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
CGAMD_FORCE_DP=1 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fid --no-roofline --no-legs 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_dp1.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_cifar -o prof -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-fid --no-legs > $R/gpurun_out/prof_${TAG}_cifar.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_dstep -o prof -- python $R/scripts/run_leg.py resnet128_dstep 5 > $R/gpurun_out/prof_${TAG}_dstep.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_fid -o prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-legs > $R/gpurun_out/prof_${TAG}_fid.log 2>&1
rm -rf /tmp/pmc_f /tmp/pmc_w
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o p -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-fid --no-legs > $R/gpurun_out/${TAG}_pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o p -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-fid --no-legs > $R/gpurun_out/${TAG}_pmc_write.log 2>&1
cd $R
python scripts/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_traffic.txt 2>&1
find gpurun_out/prof_${TAG}_* -name "*.db" -delete 2>/dev/null; find gpurun_out/prof_${TAG}_* -name "*kernel_trace.csv" -delete 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1])
print('cifar', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'])
print('fid10k', d['fid10k']['wall_s'], d['fid10k'].get('extractor_setup_s'), d['fid10k']['split_s'])
print('cpu', d['cpu_baseline']['value'])
for leg in ['resnet128_dstep','resnet128_dstep_gp','biggan128']:
    L=d.get(leg)
    if L: print(leg, L.get('ms'), L.get('tflops'), L.get('frac'), L.get('error'))
PY
cat gpurun_out/${TAG}_bench_dp1.json | cut -c1-200
head -8 gpurun_out/${TAG}_pmc_traffic.txt
