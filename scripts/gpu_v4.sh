#!/bin/bash
# Visit v4: new architectures, eval path (device-side FID assembly, padded Inception), Inception profile
TAG=${1:-v4}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_architectures_gpu.py tests/test_eval_gpu.py -m gpu -q -s --durations=6 2>&1 | tail -40 > gpurun_out/${TAG}_tests.txt
cat gpurun_out/${TAG}_tests.txt
CGAMD_INCEPTION_PAD=0 timeout 300 python scripts/prof_inception.py 64 > gpurun_out/${TAG}_inception_nopad.txt 2>&1
timeout 300 python scripts/prof_inception.py 64 > gpurun_out/${TAG}_inception.txt 2>&1
head -3 gpurun_out/${TAG}_inception_nopad.txt; head -45 gpurun_out/${TAG}_inception.txt; tail -12 gpurun_out/${TAG}_inception.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('fid10k', d['fid10k'])"
