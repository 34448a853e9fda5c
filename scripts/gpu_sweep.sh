#!/bin/bash
# Dispatch-threshold sweep on the resnet128 D-step leg and the cifar step (interleaved with the default)
TAG=${1:-sweep}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
run() {
  echo "== $*"
  env "$@" timeout 200 python scripts/run_leg.py resnet128_dstep 20 2>/dev/null | tail -1 | python -c "import json,sys; L=json.load(sys.stdin); print('   dstep ms', L['ms'], 'frac', L['frac'])"
  env "$@" timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-fid --no-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('   cifar ms', d['ms_per_step'])"
}
{
run CGAMD_X=0
run CGAMD_CONV_SK=0
run CGAMD_CONV_SK=512
run CGAMD_X=0
run CGAMD_HCONV_MIN=32
run CGAMD_HWGRAD_MIN=32
run CGAMD_HWGRAD_BLOCKS=128
run CGAMD_HWGRAD_BLOCKS=512
run CGAMD_X=0
run CGAMD_HALO_BLOCKS=256
run CGAMD_HALO_BLOCKS=1024
run CGAMD_CONV_T128_MIN=257
run CGAMD_X=0
} 2>&1 | tee gpurun_out/${TAG}.txt
