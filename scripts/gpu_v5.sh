#!/bin/bash
# Visit v5: architecture tests again (full output), SSGAN, Inception chunk size A/B on FID-10k
TAG=${1:-v5}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_architectures_gpu.py tests/test_ssgan_gpu.py tests/test_eval_gpu.py -m gpu -q -s -k "architecture or resnet_stl or ssgan or rotate or inception_features" 2>&1 | tail -150 > gpurun_out/${TAG}_tests.txt
grep -n "max / mean\|worst\|ssgan\|passed\|failed\|^E  \|Error" gpurun_out/${TAG}_tests.txt | head -60
for b in 64 256 512; do
  echo "== CGAMD_INCEPTION_BATCH=$b"
  CGAMD_INCEPTION_BATCH=$b timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('fid10k', d['fid10k']['wall_s'], d['fid10k']['split_s'], d['fid10k']['fid'])"
done 2>&1 | tee gpurun_out/${TAG}_inception_ab.txt
