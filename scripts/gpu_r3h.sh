#!/bin/bash
# round 3: full bench line with the calibration fields, forced-DP teardown x4, new kernel tests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "multi_grouped or deferred or small_all or no_small or small_" > gpurun_out/r3h_tests.txt 2>&1
tail -4 gpurun_out/r3h_tests.txt
for m in 0 1; do
  CGAMD_SCONV=$m CGAMD_TEST_REPORT=1 timeout 600 python -m pytest tests/test_modular_gan_gpu.py -x -q -m gpu -k "test_wgangp_step_resnet5" -s 2>&1 | grep -E "cos |passed|failed|Error" | sort | head -12
done
timeout 900 python bench.py > gpurun_out/r3h_bench.json 2> gpurun_out/r3h_bench.err
tail -3 gpurun_out/r3h_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3h_bench.json').read().strip().splitlines()[-1])
print('cifar', d['value'], d['ms_per_step'], 'preheat', d['preheat_s'], d['preheat_steps'], 'clocks', d['clocks'])
print('calibration', d['calibration'])
r=d['roofline']; print('roofline', r['kernel'], r['achieved'], r['frac'], r['frac_of_measured'], r['traffic'])
print('fid10k', {k:v for k,v in d['fid10k'].items() if k not in ('note','wall_definition')})
print('cpu', d['cpu_baseline'])
for leg in ['resnet128_dstep','resnet128_dstep_gp','biggan128','biggan128_bs256']:
    L=d.get(leg,{})
    print(leg, L.get('ms'), L.get('tflops'), L.get('frac'), L.get('frac_of_measured'), L.get('error'), L.get('cpu_baseline'))
PY
for i in 1 2 3 4; do
  CGAMD_FORCE_DP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2950$i RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fid --no-legs --no-roofline > gpurun_out/r3h_dp$i.json 2> gpurun_out/r3h_dp$i.err
  echo "forced-dp run $i rc=$? $(python -c "import json;d=json.loads(open('gpurun_out/r3h_dp$i.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])" 2>&1 | tail -1)"
done
