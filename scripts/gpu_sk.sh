#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
{
CGAMD_CONV_SK=256 timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gconv or stem or linear" 2>&1 | tail -4
for v in 0 256 0 256; do
  echo "== SK=$v"; CGAMD_CONV_SK=$v timeout 100 python scripts/bench_convs.py cifar 2>&1 | grep -v amdgpu | cut -c1-58
done
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fid --no-roofline"
for v in 0 256 384 0 256 384; do
  echo -n "SK=$v: "; CGAMD_CONV_SK=$v timeout 200 $B 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
} > gpurun_out/sk.txt 2>&1
cat gpurun_out/sk.txt
