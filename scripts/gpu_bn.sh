#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
CGAMD_HCONV_MIN=1 CGAMD_HWGRAD_MIN=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "test_gconv_fused_batch_norm or test_gconv_forward_adjoint_wgrad" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_modular_gan_gpu.py -m gpu -q -x -k "fused_batch_norm or train_steps_resnet_cifar or test_forward_and_gradients" 2>&1 | tail -12
timeout 300 python scripts/run_leg.py resnet128_dstep 10 2>/dev/null | tail -1 > gpurun_out/bn_dstep.json
python - <<PY
import json
L=json.load(open('gpurun_out/bn_dstep.json'))
print('dstep', L['ms'], L['tflops'], L['frac'], 'conv ms', L['conv_kernel_ms_eager'])
PY
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-fid --no-legs 2>/dev/null | tail -1 | cut -c1-200
