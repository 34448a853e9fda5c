cd $GRAFT_REPO_ROOT
T=r05_v15
for v in 0 1 0 1; do
  CGAMD_WGRAD_STREAM=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fid --no-roofline --legs resnet128_dstep,biggan128 > gpurun_out/${T}_ab$v.json 2> gpurun_out/${T}_ab$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_ab$v.json').read().strip().splitlines()[-1])
print('CGAMD_WGRAD_STREAM=$v cifar %.4f ms  dstep %.4f ms biggan128 %.3f' % (d['ms_per_step'], d['resnet128_dstep']['ms'], d['biggan128']['ms']))
PY
done | tee gpurun_out/${T}_wgrad_stream_ab.txt
