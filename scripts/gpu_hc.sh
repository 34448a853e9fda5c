#!/bin/bash
# hconv / hwgrad iteration: parity subset, microbench A/B, workgroup timeline.  usage: gpu_hc.sh TAG [wgrad_ab]
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
CGAMD_HCONV_MIN=1 CGAMD_HWGRAD_MIN=1 CGAMD_HCONV_RW_MIN=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "test_gconv_forward_adjoint_wgrad or test_gconv_gates_residual or test_stem_relu_gate" 2>&1 | tail -12 > gpurun_out/hc_${TAG}_tests.txt
cat gpurun_out/hc_${TAG}_tests.txt
timeout 300 python scripts/bench_convs.py ${SHAPES:-hc} 2>&1 | grep -v amdgpu.ids > gpurun_out/hc_${TAG}_convs.txt
cat gpurun_out/hc_${TAG}_convs.txt
if [ -n "$2" ]; then
  echo "== $2" >> gpurun_out/hc_${TAG}_convs.txt
  env $2 timeout 300 python scripts/bench_convs.py hc 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/hc_${TAG}_convs.txt
fi
