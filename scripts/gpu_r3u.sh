#!/bin/bash
# round 3: attention kernels with register-staged tiles, small-linear / thin kernels v2, tile policy
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "thin or lin_ or gemm_1x1 or attention or fast_ or c96 or halo_4x4" 2>&1 | tail -4
rm -f gpurun_out/r3u_launches.txt
CGAMD_PROF_LOG=$R/gpurun_out/r3u_launches.txt timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fid --no-roofline --legs biggan128 > gpurun_out/r3u_bench.json 2> gpurun_out/r3u_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3u_bench.json").read().strip().splitlines()[-1])
l=d["biggan128"]; print("cifar", d["ms_per_step"], "biggan ms", l["ms"], "conv eager", l["conv_kernel_ms_eager"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_big -o prof -- python $R/scripts/run_leg_eager.py biggan128 3 > $R/gpurun_out/r3u_biggan.log 2>&1
cd $R
cp $(find /tmp/p_big -name "*kernel_stats.csv" | head -1) gpurun_out/r3u_biggan_kernel_stats.csv
grep -i "attn" gpurun_out/r3u_biggan_kernel_stats.csv | cut -c1-60,200-400
