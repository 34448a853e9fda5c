#!/bin/bash
# round 3, visit 1: per-geometry conv timings of the D-step leg + an ORDERED kernel trace of it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 200 python scripts/prof_leg_shapes.py resnet128_dstep > gpurun_out/r3a_shapes_dstep.txt 2>&1
timeout 200 python scripts/prof_leg_shapes.py cifar > gpurun_out/r3a_shapes_cifar.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $R/scripts/run_leg.py resnet128_dstep 3 > $R/gpurun_out/r3a_kt.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last replayed step = the last N rows between two adam launches
idx = [i for i, r in enumerate(rows) if 'adam_multi' in r['Kernel_Name']]
lo, hi = idx[-2] + 1, idx[-1] + 1
with open('gpurun_out/r3a_dstep_order.txt', 'w') as o:
    t0 = int(rows[lo]['Start_Timestamp'])
    for r in rows[lo:hi]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        o.write('%9.1f %8.1f  %s  grid=%s wg=%s\n' % ((s - t0) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:110], r.get('Grid_Size_X', r.get('Grid_Size')), r.get('Workgroup_Size_X', r.get('Workgroup_Size'))))
print('launches in last step', hi - lo)
PY
head -50 gpurun_out/r3a_shapes_dstep.txt
