#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/scripts/pmc_convs.py > $R/gpurun_out/trace_convs.log 2>&1
python - <<'PY' > $R/gpurun_out/trace_convs.txt
import csv, glob
rows = []
for f in glob.glob('/tmp/tr/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for r in rows:
    n = r['Kernel_Name']
    if 'at::native' in n: continue
    print("%8.1f us  grid %-8s lds %-7s %s" % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size_X', r.get('Grid_Size','')), r.get('LDS_Block_Size',''), n[:90]))
PY
tail -40 $R/gpurun_out/trace_convs.txt
