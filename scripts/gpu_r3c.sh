#!/bin/bash
# round 3: weight-gradient dispatch alternatives on the small-map shapes (one box visit)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
CGAMD_SCONV=0 CGAMD_NO_HALO_WGRAD=1 timeout 300 python scripts/check_small_conv.py > gpurun_out/r3c_nohalo.txt 2>&1
CGAMD_SCONV=0 CGAMD_HALO_BLOCKS=256 timeout 300 python scripts/check_small_conv.py > gpurun_out/r3c_hb256.txt 2>&1
CGAMD_SCONV=0 CGAMD_HALO_BLOCKS=128 timeout 300 python scripts/check_small_conv.py > gpurun_out/r3c_hb128.txt 2>&1
for f in nohalo hb256 hb128; do echo "== $f"; awk '{print $1, $(NF-1), $NF}' gpurun_out/r3c_$f.txt | tail -15; done
