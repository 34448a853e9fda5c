cd $GRAFT_REPO_ROOT
T=r05_v12
timeout 600 python -m pytest -q -m gpu tests/test_kernels_gpu.py -x -k "pooled_head" 2>&1 | tail -12 > gpurun_out/${T}_t_kern.txt
timeout 900 python -m pytest -q -m gpu tests/test_modular_gan_gpu.py -x -k "forward_and_gradients and not baseline_batch or wgangp_step or biggan_forward or captured" 2>&1 | tail -12 > gpurun_out/${T}_t_gan.txt
timeout 600 python -m pytest -q -m gpu tests/test_ssgan_gpu.py tests/test_s3gan_gpu.py tests/test_tf_checkpoint.py tests/test_architectures_gpu.py -x -k "not resnet30" 2>&1 | tail -12 > gpurun_out/${T}_t_misc.txt
for f in gpurun_out/${T}_t_*.txt; do echo "== $f"; tail -n 8 $f; done
for v in 0 1 0 1; do
  CGAMD_FUSED_HEAD=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fid --no-roofline --legs resnet128_dstep > gpurun_out/${T}_ab_head$v.json 2> gpurun_out/${T}_ab_head$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_ab_head$v.json').read().strip().splitlines()[-1])
print('CGAMD_FUSED_HEAD=$v cifar %.4f ms  dstep %.4f ms' % (d['ms_per_step'], d['resnet128_dstep']['ms']))
PY
done | tee gpurun_out/${T}_fused_head_ab.txt
