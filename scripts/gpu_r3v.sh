#!/bin/bash
# round 3: split block-Jacobi rounds (A/B against the fused round kernel), device-side synthetic
# extractor weights: FID-10k wall
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_eval_gpu.py -x -q -m gpu -k "fid or eval or inception or jacobi or syevj" 2>&1 | tail -4
for sp in 0 1; do export CGAMD_JACOBI_GRAPH=$sp;
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-legs > gpurun_out/r3v_bench_$sp.json 2> gpurun_out/r3v_bench_$sp.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3v_bench_$sp.json").read().strip().splitlines()[-1])
print("graph $sp", json.dumps(d["fid10k"])[:420])
PY
done
