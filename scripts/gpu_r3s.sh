#!/bin/bash
# round 3: per-launch geometry log of the BigGAN-128 leg
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; rm -f gpurun_out/r3s_biggan_launches.txt
CGAMD_PROF_LOG=$R/gpurun_out/r3s_biggan_launches.txt timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fid --no-roofline --legs biggan128 > gpurun_out/r3s_bench.json 2> gpurun_out/r3s_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3s_bench.json").read().strip().splitlines()[-1])
l=d["biggan128"]; print("biggan ms", l["ms"], "conv eager", l["conv_kernel_ms_eager"])
PY
wc -l gpurun_out/r3s_biggan_launches.txt
