#!/bin/bash
# round 3: LDS-DMA from inline asm in the weight-gradient kernels: correctness + A/B in one visit
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gconv or wgrad or pool or stem or deferred" > gpurun_out/r3e_tests.txt 2>&1
tail -4 gpurun_out/r3e_tests.txt
for m in 1 0 1 0; do
  CGAMD_ASM_DMA=$m timeout 300 python scripts/bench_convs.py resnet128 2>&1 | grep -v amdgpu.ids > gpurun_out/r3e_convs_asm$m.txt
done
paste <(awk '{print $1, $NF}' gpurun_out/r3e_convs_asm1.txt) <(awk '{print $NF}' gpurun_out/r3e_convs_asm0.txt)
for m in 1 0; do
  CGAMD_ASM_DMA=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fid > gpurun_out/r3e_bench_asm$m.json 2> gpurun_out/r3e_bench_asm$m.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r3e_bench_asm$m.json').read().strip().splitlines()[-1])
print('asm_dma=$m: cifar ms', d['ms_per_step'], 'dstep', d.get('resnet128_dstep',{}).get('ms'), 'dstep_gp', d.get('resnet128_dstep_gp',{}).get('ms'), 'biggan', d.get('biggan128',{}).get('ms'))
PY
done
