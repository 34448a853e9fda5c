#!/bin/bash
# Round 2, visit A: baseline of the new bench legs + where the resnet128 D-step spends its time.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 3 2>gpurun_out/r2a_bench.err | tail -1 > gpurun_out/r2a_bench.json
echo "bench rc=$?"
timeout 300 python scripts/prof_leg_shapes.py resnet128_dstep > gpurun_out/r2a_shapes_dstep.txt 2>&1
timeout 300 python scripts/prof_leg_shapes.py resnet128_dstep_gp > gpurun_out/r2a_shapes_dstep_gp.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2a_prof_dstep -o prof -- python $R/scripts/run_leg.py resnet128_dstep 5 > $R/gpurun_out/r2a_prof_dstep.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/scripts/bench_convs.py resnet128 > $R/gpurun_out/r2a_convs_r128.log 2>&1
python - <<'PY' > $R/gpurun_out/r2a_convs_r128_trace.txt
import csv, glob, collections
rows = []
for f in glob.glob('/tmp/tr/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last graph replay of every shape: print distinct (kernel, grid) with the median duration
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if 'at::native' in n: continue
    k = (n[:100], r.get('Grid_Size_X', r.get('Grid_Size','')), r.get('Grid_Size_Y',''), r.get('LDS_Block_Size',''), r.get('VGPR_Count', ''))
    agg.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in agg.items():
    v.sort()
    print("%8.1f us (n=%3d) grid %-8s y %-4s lds %-7s vgpr %-4s %s" % (v[len(v)//2], len(v), k[1], k[2], k[3], k[4], k[0]))
PY
cd $R
find gpurun_out/r2a_prof_dstep -name "*.db" -delete 2>/dev/null
find gpurun_out/r2a_prof_dstep -name "*kernel_trace.csv" -delete 2>/dev/null
cut -c1-600 gpurun_out/r2a_bench.json; tail -3 gpurun_out/r2a_bench.err
head -30 gpurun_out/r2a_shapes_dstep.txt
