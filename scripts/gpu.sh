#!/bin/bash
# ONE launcher for every GPU visit:   gpurun --timeout T -- 'bash scripts/gpu.sh <task> [args]'
# Everything is written under gpurun_out/ (scratch, merged back by gpurun); copy what is worth
# keeping to profiles/.  TAG prefixes the output files (default: the task name).
#
#   full [TAG]                      complete `-m gpu` suite (log kept) + __graft_entry__.smoke()
#   tests [TAG] -- PYTEST_ARGS...   pytest tests/ -m gpu with the given arguments (e.g. -k attention)
#   bench [TAG] -- BENCH_ARGS...    python bench.py ... -> gpurun_out/TAG_bench.json (one JSON line)
#   launches LEG [TAG]              per-launch geometry log of a leg's convolutions (CGAMD_PROF_LOG):
#                                   LEG = resnet128_dstep | resnet128_dstep_gp | biggan128 | biggan128_bs256
#   stats WORKLOAD [TAG]            rocprofv3 --kernel-trace --stats summary (csv):
#                                   WORKLOAD = cifar | fid | resnet128_dstep | resnet128_dstep_gp | biggan128
#   traffic WORKLOAD                the two --pmc passes (FETCH_SIZE, WRITE_SIZE; own runs, no other
#                                   trace domain) + scripts/pmc_traffic.py -> gpurun_out/r06_pmc_traffic.json
#                                   WORKLOAD = cifar | resnet128_dstep
#   ab VAR V1,V2,... LEG [TAG]      the same build under VAR=V1, VAR=V2, ... on one box (boxes of the
#                                   pool differ by +-15 % in clocks): bench.py --legs LEG, prints the
#                                   headline step, the leg's step and its per-family kernel times
#   ablib LEGS [TAG]                lib/libcgamd_prev.so (a build of an earlier commit, made by hand) against
#                                   lib/libcgamd.so through CGAMD_LIB_PATH, alternating twice: bench.py --legs LEGS
#   dp [TAG]                        CGAMD_FORCE_DP=1: the data-parallel path on a one-rank RCCL group
#                                   (bucket, all-reduce captured in the hipGraph, bucketed overlap on / off)
#   final [TAG]                     scripts/visit_final.sh: bench (all legs) FIRST, rocprofv3 stats, PMC passes, dp, suite LAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
mkdir -p gpurun_out
task=$1; shift
split_args() {   # TAG [--] rest...
  TAG=$1
  if [ "$TAG" = "--" ] || [ -z "$TAG" ]; then TAG=$task; else shift; fi
  [ "$1" = "--" ] && shift
  REST=("$@")
}
run_full() {
  timeout 2400 python -m pytest tests/ -q -m gpu --durations=15 > gpurun_out/$1_tests.txt 2>&1
  tail -25 gpurun_out/$1_tests.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
}
leg_summary() {   # file.json LEG
  python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
out = "cifar %.3f ms/step" % d["ms_per_step"]
leg = d.get(sys.argv[2]) if len(sys.argv) > 2 else None
if leg and "ms" in leg:
    out += " | %s %.3f ms | %s" % (sys.argv[2], leg["ms"], {k: round(v["ms_per_step"], 3)
                                                          for k, v in leg.get("kernels", {}).items()})
if d.get("fid10k"):
    out += " | fid10k %s" % {k: d["fid10k"].get(k) for k in ("wall_s", "split_s")}
print(out)
PY
}
stats_cmd() {   # WORKLOAD -> command line profiled
  case $1 in
    cifar) echo "python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-fid --no-legs --no-calibration" ;;
    fid) echo "python $R/bench.py --steps 2 --warmup 1 --preheat-s 0 --no-cpu-baseline --no-roofline --no-legs --no-calibration" ;;
    *) echo "python $R/scripts/run_leg_eager.py $1 3" ;;
  esac
}
run_stats() {   # WORKLOAD TAG
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_$1 &&
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$1 -o prof -- $(stats_cmd $1) \
      > "$R/gpurun_out/$2_$1.log" 2>&1 )
  cp "$(find /tmp/p_$1 -name '*kernel_stats.csv' | head -1)" gpurun_out/$2_$1_kernel_stats.csv
  head -12 gpurun_out/$2_$1_kernel_stats.csv | cut -c1-150
}
run_dp() {
  for ov in 0 1; do
    CGAMD_FORCE_DP=1 CGAMD_DP_OVERLAP=$ov CGAMD_DP_BUCKET_MIN_MB=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2956$ov \
      RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline \
      --no-fid --no-legs --no-roofline > gpurun_out/$1_dp_ov$ov.json 2> gpurun_out/$1_dp_ov$ov.err
    echo "forced DP, overlap $ov: $(leg_summary gpurun_out/$1_dp_ov$ov.json)"
  done
}
case $task in
  full) split_args "$@"; run_full $TAG ;;
  tests) split_args "$@"; timeout 2400 python -m pytest -q -m gpu "${REST[@]}" 2>&1 | tee gpurun_out/${TAG}_tests.txt | tail -15 ;;
  bench) split_args "$@"
    timeout 1500 python bench.py "${REST[@]}" > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
    tail -c 1500 gpurun_out/${TAG}_bench.json ;;
  launches) LEG=$1; TAG=${2:-launches}; rm -f gpurun_out/${TAG}_$LEG.txt
    CGAMD_PROF_LOG=$R/gpurun_out/${TAG}_$LEG.txt timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline \
      --no-fid --no-roofline --legs $LEG > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
    leg_summary gpurun_out/${TAG}_bench.json $LEG; wc -l gpurun_out/${TAG}_$LEG.txt ;;
  stats) run_stats $1 ${2:-stats} ;;
  traffic) W=$1
    case $W in cifar) CMD="python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-fid --no-legs --preheat-s 0 --no-calibration" ;;
               *) CMD="python $R/scripts/run_leg_eager.py $W 2" ;; esac
    ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pf_$W /tmp/pw_$W &&
      timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$W -o p -- $CMD > "$R/gpurun_out/traffic_pf_$W.log" 2>&1 &&
      timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw_$W -o p -- $CMD > "$R/gpurun_out/traffic_pw_$W.log" 2>&1 )
    python scripts/pmc_traffic.py /tmp/pf_$W /tmp/pw_$W gpurun_out/r06_pmc_traffic.json $W | head -14 ;;
  ab) VAR=$1; VALS=$2; LEG=$3; TAG=${4:-ab}
    for v in ${VALS//,/ }; do
      env $VAR=$v timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fid --no-roofline --legs $LEG \
        > gpurun_out/${TAG}_${VAR}_$v.json 2> gpurun_out/${TAG}_${VAR}_$v.err
      echo "$VAR=$v: $(leg_summary gpurun_out/${TAG}_${VAR}_$v.json $LEG)"
    done ;;
  ablib) LEGS=$1; TAG=${2:-ablib}   # previous vs current library, alternating (A B A B) on one box
    for rep in 1 2; do
      for which in prev cur; do
        LIBP=$R/compare_gan_amd/lib/libcgamd.so; [ $which = prev ] && LIBP=$R/compare_gan_amd/lib/libcgamd_prev.so
        CGAMD_LIB_PATH=$LIBP timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fid --no-roofline \
          --legs $LEGS > gpurun_out/${TAG}_${which}$rep.json 2> gpurun_out/${TAG}_${which}$rep.err
        for L in ${LEGS//,/ }; do echo "$which $rep: $(leg_summary gpurun_out/${TAG}_${which}$rep.json $L)"; done
      done
    done ;;
  sq) TAG=$1; SHAPE=$2; KINDS=${3:-fwd}; shift 3   # SQ counter passes over one conv shape; extra args: VAR=V settings
    ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sq_$TAG && i=0 &&
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
                 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
                 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
        i=$((i+1))
        env "$@" timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/sq_$TAG/p$i -o p -- python $R/scripts/pmc_one.py run $SHAPE $KINDS > "$R/gpurun_out/${TAG}_sq_p$i.log" 2>&1
      done )
    echo "# shape $SHAPE kinds $KINDS settings $*" > gpurun_out/${TAG}_sq.txt
    python scripts/pmc_one.py agg /tmp/sq_$TAG >> gpurun_out/${TAG}_sq.txt; cat gpurun_out/${TAG}_sq.txt ;;
  dp) run_dp ${1:-dp} ;;
  final) exec bash "$R/scripts/visit_final.sh" "${1:-final}" ;;   # bench first, profiles, suite last
  *) echo "unknown task '$task' (see the header of scripts/gpu.sh)"; exit 2 ;;
esac
