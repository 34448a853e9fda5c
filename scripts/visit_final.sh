#!/bin/bash
# The closing measurement of a round, on the FINAL tree: bench with every
# leg FIRST, then the rocprofv3 summaries, the two PMC passes, the data-parallel dry runs, and the
# complete -m gpu suite SEQUENTIALLY last (a slow suite can then cost at most itself).
#   gpurun --timeout 3000 -- 'bash scripts/visit_final.sh TAG'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
T=${1:-r06_final}
mkdir -p gpurun_out
echo "== bench (all legs)"; date +%s
timeout 1200 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 1200 gpurun_out/${T}_bench.json
echo "== rocprofv3 --kernel-trace --stats"; date +%s
stats() {  # NAME CMD...
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_$name &&
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -o prof -- "$@" \
      > "$R/gpurun_out/${T}_$name.log" 2>&1 )
  cp "$(find /tmp/p_$name -name '*kernel_stats.csv' | head -1)" gpurun_out/${T}_${name}_kernel_stats.csv
  head -4 gpurun_out/${T}_${name}_kernel_stats.csv | cut -c1-140
}
stats cifar python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-fid --no-legs --no-calibration
stats fid python $R/bench.py --steps 2 --warmup 1 --preheat-s 0 --no-cpu-baseline --no-roofline --no-legs --no-calibration
stats resnet128_dstep python $R/scripts/run_leg_eager.py resnet128_dstep 3
stats biggan128_bs256 python $R/scripts/run_leg_eager.py biggan128_bs256 3
echo "== PMC traffic (FETCH_SIZE / WRITE_SIZE, own passes)"; date +%s
for W in cifar resnet128_dstep biggan128_bs256; do
  case $W in cifar) CMD="python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-fid --no-legs --preheat-s 0 --no-calibration" ;;
             *) CMD="python $R/scripts/run_leg_eager.py $W 2" ;; esac
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pf_$W /tmp/pw_$W &&
    timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$W -o p -- $CMD > "$R/gpurun_out/${T}_pf_$W.log" 2>&1 &&
    timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw_$W -o p -- $CMD > "$R/gpurun_out/${T}_pw_$W.log" 2>&1 )
  python scripts/pmc_traffic.py /tmp/pf_$W /tmp/pw_$W gpurun_out/r06_pmc_traffic.json $W | head -12
done
# the committed traffic table bench.py reads must not be older than the library
if [ -f profiles/r06_pmc_traffic.json ] && [ profiles/r06_pmc_traffic.json -ot compare_gan_amd/lib/libcgamd.so ]; then
  echo "NOTE: profiles/r06_pmc_traffic.json is older than libcgamd.so -- copy gpurun_out/r06_pmc_traffic.json over it after this visit"
fi
echo "== forced DP (one-rank RCCL group, overlap off / on)"; date +%s
for ov in 0 1; do
  CGAMD_FORCE_DP=1 CGAMD_DP_OVERLAP=$ov CGAMD_DP_BUCKET_MIN_MB=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2956$ov \
    RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline \
    --no-fid --no-legs --no-roofline > gpurun_out/${T}_dp_ov$ov.json 2> gpurun_out/${T}_dp_ov$ov.err
  python -c "import json;d=json.loads(open('gpurun_out/${T}_dp_ov$ov.json').read().strip().splitlines()[-1]);print('forced DP overlap $ov: %.3f ms/step' % d['ms_per_step'])"
done
echo "== smoke + full -m gpu suite (sequential)"; date +%s
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
timeout 2400 python -m pytest tests/ -q -m gpu --durations=15 > gpurun_out/${T}_tests.txt 2>&1
tail -25 gpurun_out/${T}_tests.txt
date +%s
