#!/bin/bash
# Round-end GPU visit: full parity suite, smoke(), the bench line (all legs), forced-DP bench,
# rocprofv3 kernel stats, the two PMC passes for HBM traffic, BigGAN-128 bench.  Usage: gpu_final.sh TAG
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/tests_$TAG.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > gpurun_out/smoke_$TAG.log
timeout 400 python bench.py 2>&1 | tail -1 > gpurun_out/bench_$TAG.json
CGAMD_FORCE_DP=1 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fid --no-roofline 2>&1 | tail -1 > gpurun_out/bench_dp1_$TAG.json
timeout 300 python bench.py --config biggan_imagenet128.gin --batch-per-gpu 32 --steps 8 --warmup 2 --no-cpu-baseline --no-fid 2>&1 | tail -1 > gpurun_out/bench_biggan_$TAG.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-fid > $R/gpurun_out/prof_$TAG.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o p -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-fid > $R/gpurun_out/pmc_fetch_$TAG.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o p -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-fid > $R/gpurun_out/pmc_write_$TAG.log 2>&1
cd $R
python scripts/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w gpurun_out/pmc_traffic_$TAG.json > gpurun_out/pmc_traffic_$TAG.txt 2>&1
rm -f gpurun_out/prof_$TAG/*kernel_trace.csv
cat gpurun_out/tests_$TAG.log gpurun_out/smoke_$TAG.log; cut -c1-330 gpurun_out/bench_$TAG.json; cut -c1-200 gpurun_out/bench_dp1_$TAG.json; cut -c1-330 gpurun_out/bench_biggan_$TAG.json; cat gpurun_out/pmc_traffic_$TAG.txt
