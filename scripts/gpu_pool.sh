#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
CGAMD_HCONV_MIN=1 CGAMD_HWGRAD_MIN=1 CGAMD_HCONV_RW_MIN=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "test_conv_pool_fused or test_gconv_fused_batch_norm" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_modular_gan_gpu.py -m gpu -q -x -k "test_forward_and_gradients or wgangp or train_steps" 2>&1 | tail -8
for v in 0 1 0 1; do
  echo "== CGAMD_FUSED_POOL=$v"
  CGAMD_FUSED_POOL=$v timeout 300 python scripts/run_leg.py resnet128_dstep 20 2>/dev/null | tail -1 | python -c "import json,sys; L=json.load(sys.stdin); print('dstep', L['ms'], L['frac'], 'conv', L['conv_kernel_ms_eager'])"
  CGAMD_FUSED_POOL=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-fid --no-legs 2>/dev/null | tail -1 | cut -c1-140
done
