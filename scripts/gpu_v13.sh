#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_modular_gan_gpu.py tests/test_modular_gan_matrix_gpu.py tests/test_data_parallel_gpu.py -m gpu -q -x -k "forward_and_gradients or captured or wgangp or penalties or biggan_forward or single_training_step_arch or force_dp" 2>&1 | tail -5 | tee gpurun_out/v13_tests.txt
for rep in 1 2; do for v in 0 1; do
  echo "== CGAMD_FORK=$v"
  CGAMD_FORK=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-fid --no-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('   cifar ms', d['ms_per_step'])"
  CGAMD_FORK=$v timeout 200 python scripts/run_leg.py resnet128_dstep 20 2>/dev/null | tail -1 | python -c "import json,sys; L=json.load(sys.stdin); print('   dstep ms', L['ms'], 'frac', L['frac'])"
done; done 2>&1 | tee gpurun_out/v13_ab.txt
