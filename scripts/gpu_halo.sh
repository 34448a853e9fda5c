#!/bin/bash
# halo-wgrad check: conv op tests, conv microbench with and without the halo weight-gradient kernel, short bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gconv or conv or stem" 2>&1 | tail -12 > gpurun_out/tests_halo.log
for w in cifar resnet128; do
  timeout 150 python scripts/bench_convs.py $w > gpurun_out/convs_halo_$w.txt 2>&1
  CGAMD_NO_HALO_WGRAD=1 timeout 150 python scripts/bench_convs.py $w > gpurun_out/convs_nohalo_$w.txt 2>&1
done
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fid 2>&1 | tail -1 > gpurun_out/bench_halo.log
tail -5 gpurun_out/tests_halo.log; cut -c1-400 gpurun_out/bench_halo.log
paste <(awk '{print $1, $NF-0, $(NF-1)}' gpurun_out/convs_halo_cifar.txt) <(awk '{print $(NF-1)}' gpurun_out/convs_nohalo_cifar.txt) | head -20
