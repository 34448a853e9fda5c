#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for e in "CGAMD_X=0" "CGAMD_FUSED_POOL=0" "CGAMD_WSTEM=0" "CGAMD_FUSED_POOL=0 CGAMD_WSTEM=0" "CGAMD_HCONV=0 CGAMD_HWGRAD=0"; do
  echo "== $e"
  env $e timeout 200 python scripts/debug_gp.py 2>&1 | grep -v amdgpu.ids | head -24
done | tee gpurun_out/gp_debug.txt
