cd $GRAFT_REPO_ROOT
T=r05_v17
timeout 600 python -m pytest -q -m gpu tests/test_kernels_gpu.py -x -k "sigma_folded or attention" 2>&1 | tail -6 > gpurun_out/${T}_t_kern.txt
timeout 900 python -m pytest -q -m gpu tests/test_modular_gan_gpu.py -x -k "biggan and not 512 and not 256px" 2>&1 | tail -8 > gpurun_out/${T}_t_gan.txt
timeout 600 python -m pytest -q -m gpu tests/test_eval_gpu.py tests/test_s3gan_gpu.py -x 2>&1 | tail -5 > gpurun_out/${T}_t_misc.txt
for f in gpurun_out/${T}_t_*.txt; do echo "== $f"; tail -n 6 $f; done
for v in 0 1 0 1; do
  CGAMD_FOLD_SIGMA=$v timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fid --no-roofline --legs biggan128,biggan128_bs256 > gpurun_out/${T}_ab$v.json 2> gpurun_out/${T}_ab$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_ab$v.json').read().strip().splitlines()[-1])
print('CGAMD_FOLD_SIGMA=$v biggan128 %.3f ms  biggan128_bs256 %.3f ms' % (d['biggan128']['ms'], d['biggan128_bs256']['ms']))
PY
done | tee gpurun_out/${T}_fold_sigma_ab.txt
