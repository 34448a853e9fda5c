#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"
timeout 200 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pmc_halo -o p -- python $R/scripts/pmc_convs.py > $R/gpurun_out/pmc_halo.log 2>&1
CGAMD_NO_HALO_WGRAD=1 timeout 200 rocprofv3 --pmc $PMC --output-format csv -d /tmp/pmc_tap -o p -- python $R/scripts/pmc_convs.py > $R/gpurun_out/pmc_tap.log 2>&1
python $R/scripts/pmc_convs.py agg /tmp/pmc_halo > $R/gpurun_out/pmc_halo_agg.txt 2>&1
python $R/scripts/pmc_convs.py agg /tmp/pmc_tap > $R/gpurun_out/pmc_tap_agg.txt 2>&1
tail -3 $R/gpurun_out/pmc_halo.log; wc -l $R/gpurun_out/pmc_*_agg.txt
