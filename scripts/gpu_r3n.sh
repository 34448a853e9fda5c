#!/bin/bash
# round 3: l2 penalty (multi-tensor), BigGAN 256 px, sharded-eval refactor (single rank), bucketed
# all-reduce on the one-rank RCCL group + its bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_modular_gan_gpu.py -x -q -m gpu -k "penalties or 256px" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_eval_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python -m pytest tests/test_data_parallel_gpu.py -x -q -m gpu -k "force_dp" 2>&1 | tail -8
for ov in 0 1; do
  CGAMD_FORCE_DP=1 CGAMD_DP_OVERLAP=$ov CGAMD_DP_BUCKET_MIN_MB=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2956$ov RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fid --no-legs --no-roofline > gpurun_out/r3n_dp_ov$ov.json 2> gpurun_out/r3n_dp_ov$ov.err
  tail -1 gpurun_out/r3n_dp_ov$ov.json | cut -c1-300
done
