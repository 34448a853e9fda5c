cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r05_v3
timeout 900 python -m pytest -q -m gpu tests/test_eval_gpu.py -x -k "sink or nan or sampling_speed or evaluate_gan_small" 2>&1 | tail -8 > gpurun_out/${T}_t_eval.txt
timeout 900 python -m pytest -q -m gpu tests/test_kernels_gpu.py -x -k "deferred or full_size" 2>&1 | tail -8 > gpurun_out/${T}_t_kern.txt
timeout 900 python -m pytest -q -m gpu tests/test_data_parallel_gpu.py -x -k "bench_entry" 2>&1 | tail -15 > gpurun_out/${T}_t_dp.txt
timeout 900 python -m pytest -q -m gpu tests/test_modular_gan_gpu.py -x -k "cifar or captured" 2>&1 | tail -8 > gpurun_out/${T}_t_gan.txt
tail -4 gpurun_out/${T}_t_*.txt
# per-launch geometry log of the D-step at HEAD
rm -f gpurun_out/${T}_launch_dstep.txt
CGAMD_PROF_LOG=$PWD/gpurun_out/${T}_launch_dstep.txt timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fid --no-roofline --legs resnet128_dstep > gpurun_out/${T}_bench_dstep.json 2> gpurun_out/${T}_bench_dstep.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_v3_bench_dstep.json').read().strip().splitlines()[-1])
print('cifar', d['ms_per_step'], 'dstep', d['resnet128_dstep']['ms'], d['resnet128_dstep']['frac'])
PY
# rocprof stats eager: dstep, biggan bs256
for w in resnet128_dstep biggan128_bs256; do
 ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_$w && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$w -o prof -- python $GRAFT_REPO_ROOT/scripts/run_leg_eager.py $w 3 > $GRAFT_REPO_ROOT/gpurun_out/${T}_$w.log 2>&1 )
 cp "$(find /tmp/p_$w -name '*kernel_stats.csv' | head -1)" gpurun_out/${T}_${w}_kernel_stats.csv
done
head -5 gpurun_out/${T}_biggan128_bs256_kernel_stats.csv | cut -c1-160
