#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_modular_gan_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "biggan_forward or biggan_full_width or attention or captured_step" 2>&1 | tail -5 | tee gpurun_out/v14_tests.txt
for v in 0 1 0 1; do
  echo "== CGAMD_PAD_ATTENTION=$v"
  CGAMD_PAD_ATTENTION=$v timeout 300 python scripts/run_leg.py biggan128 4 2>/dev/null | tail -1 | python -c "
import json,sys; L=json.load(sys.stdin); print('   biggan ms', L['ms'], 'frac', L['frac'], 'useful TFLOP', L['useful_tflop_counted'])
for k,v in list(L['kernels'].items())[:12]: print('      %-30s %7.3f ms %7.1f us %7.1f TF x%d' % (k, v['ms_per_step'], v['avg_launch_us'], v['tflops'], v['launches_per_step']))"
done 2>&1 | tee gpurun_out/v14_ab.txt
