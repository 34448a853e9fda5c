#!/bin/bash
# Visit v2: tests of the statistics groups / deferred reductions / joint generator forward, then
# interleaved A/B of the two switches on the cifar step and the resnet128 D-step.  usage: gpu_v2.sh TAG
TAG=${1:-v2}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "statistics_groups or deferred or fused_batch_norm or test_batch_norm" 2>&1 | tail -12 > gpurun_out/${TAG}_tests_k.txt
cat gpurun_out/${TAG}_tests_k.txt
timeout 900 python -m pytest tests/test_modular_gan_gpu.py -m gpu -q -x -s -k "batched_generator or joint_gen or deferred or train_steps or captured_step or fused_batch_norm" 2>&1 | tail -25 > gpurun_out/${TAG}_tests_m.txt
cat gpurun_out/${TAG}_tests_m.txt
for rep in 1 2; do
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  echo "== rep $rep CGAMD_JOINT_G=$1 CGAMD_DEFER_REDUCE=$2"
  CGAMD_JOINT_G=$1 CGAMD_DEFER_REDUCE=$2 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-fid --no-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('cifar ms', d['ms_per_step'], 'img/s', d['value'])"
done
done 2>&1 | tee gpurun_out/${TAG}_ab.txt
for rep in 1 2; do
for v in 0 1; do
  echo "== rep $rep dstep CGAMD_DEFER_REDUCE=$v"
  CGAMD_DEFER_REDUCE=$v timeout 300 python scripts/run_leg.py resnet128_dstep 20 2>/dev/null | tail -1 | python -c "import json,sys; L=json.load(sys.stdin); print('dstep ms', L['ms'], 'frac', L['frac'])"
done
done 2>&1 | tee -a gpurun_out/${TAG}_ab.txt
