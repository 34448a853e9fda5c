#!/bin/bash
# PMC passes (SQ counters) over one conv shape.  usage: gpu_pmc.sh TAG SHAPE KINDS
TAG=$1; SHAPE=$2; KINDS=${3:-fwd,wgrad}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
rm -rf /tmp/pmc_a /tmp/pmc_b /tmp/pmc_c
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d /tmp/pmc_a -o p -- python $R/scripts/pmc_one.py run $SHAPE $KINDS > $R/gpurun_out/pmc_${TAG}_a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d /tmp/pmc_b -o p -- python $R/scripts/pmc_one.py run $SHAPE $KINDS > $R/gpurun_out/pmc_${TAG}_b.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_c -o p -- python $R/scripts/pmc_one.py run $SHAPE $KINDS > $R/gpurun_out/pmc_${TAG}_c.log 2>&1
cd $R
( echo "# shape $SHAPE kinds $KINDS"; python scripts/pmc_one.py agg /tmp/pmc_a; python scripts/pmc_one.py agg /tmp/pmc_b; python scripts/pmc_one.py agg /tmp/pmc_c ) > gpurun_out/pmc_${TAG}.txt 2>&1
tail -3 gpurun_out/pmc_${TAG}_a.log gpurun_out/pmc_${TAG}_c.log | cut -c1-200
cat gpurun_out/pmc_${TAG}.txt
