#!/bin/bash
# round 3: layer-norm tests; PMC traffic (FETCH_SIZE / WRITE_SIZE passes) of the cifar step and the
# D-step leg -> profiles/r03_pmc_traffic.json
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layer_norm" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_* /tmp/pw_*
CIFAR="python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-fid --no-legs --preheat-s 0"
DSTEP="python $R/scripts/run_leg_eager.py resnet128_dstep 2"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf_cifar -o p -- $CIFAR > $R/gpurun_out/r3m_pf_cifar.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw_cifar -o p -- $CIFAR > $R/gpurun_out/r3m_pw_cifar.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf_dstep -o p -- $DSTEP > $R/gpurun_out/r3m_pf_dstep.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw_dstep -o p -- $DSTEP > $R/gpurun_out/r3m_pw_dstep.log 2>&1
cd $R
rm -f gpurun_out/r03_pmc_traffic.json
python scripts/pmc_traffic.py /tmp/pf_cifar /tmp/pw_cifar gpurun_out/r03_pmc_traffic.json cifar | head -12
python scripts/pmc_traffic.py /tmp/pf_dstep /tmp/pw_dstep gpurun_out/r03_pmc_traffic.json resnet128_dstep | head -12
tail -2 gpurun_out/r3m_pf_dstep.log | cut -c1-200
