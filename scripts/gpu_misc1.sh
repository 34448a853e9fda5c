#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_modular_gan_gpu.py -m gpu -q -x -k "wgangp or penalty" 2>&1 | tail -5
timeout 300 python scripts/prof_leg_shapes.py resnet128_dstep_gp 2>&1 | grep -v amdgpu.ids > gpurun_out/m1_shapes_gp.txt
head -40 gpurun_out/m1_shapes_gp.txt
timeout 300 python scripts/prof_leg_shapes.py biggan128 32 2>&1 | grep -v amdgpu.ids > gpurun_out/m1_shapes_biggan.txt
head -70 gpurun_out/m1_shapes_biggan.txt
