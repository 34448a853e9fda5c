#!/bin/bash
# round 2: KID / eval tests, data-parallel product tests, one-rank RCCL bench A/B
cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_data_parallel_gpu.py tests/test_eval_gpu.py -x -q 2>&1 | grep -v "frame #" > gpurun_out/dp1_tests.log
tail -25 gpurun_out/dp1_tests.log
for ov in auto 1; do
  CGAMD_FORCE_DP=1 CGAMD_DP_OVERLAP=$ov MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 timeout 300 python bench.py --steps 30 --warmup 5 --no-legs --no-cpu-baseline > gpurun_out/dp1_bench_force_ov$ov.json 2> gpurun_out/dp1_bench_force_ov$ov.err
  echo "force_dp overlap=$ov exit $?"; cut -c1-200 gpurun_out/dp1_bench_force_ov$ov.json
done
timeout 300 python bench.py --steps 30 --warmup 5 --no-legs --no-cpu-baseline > gpurun_out/dp1_bench_single.json 2> gpurun_out/dp1_bench_single.err
echo "single exit $?"; cut -c1-200 gpurun_out/dp1_bench_single.json
