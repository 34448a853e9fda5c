cd $GRAFT_REPO_ROOT
T=r05_v14
timeout 600 python -m pytest -q -m gpu tests/test_kernels_gpu.py -x -k "pooled_head" 2>&1 | tail -3
for v in 0 1 0 1; do
  CGAMD_FUSED_HEAD=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fid --no-roofline --legs resnet128_dstep > gpurun_out/${T}_ab_head$v.json 2> gpurun_out/${T}_ab_head$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_ab_head$v.json').read().strip().splitlines()[-1])
print('CGAMD_FUSED_HEAD=$v cifar %.4f ms  dstep %.4f ms' % (d['ms_per_step'], d['resnet128_dstep']['ms']))
PY
done | tee gpurun_out/${T}_fused_head_ab.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_tl && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --preheat-s 0 --no-cpu-baseline --no-roofline --no-fid --no-legs > $GRAFT_REPO_ROOT/gpurun_out/${T}_tl.log 2>&1 )
F=$(find /tmp/p_tl -name '*kernel_trace.csv' | head -1)
python scripts/graph_timeline.py $F 470 | tee gpurun_out/${T}_cifar_graph_timeline.txt
