#!/bin/bash
# interleaved A/B of builds of the library in ONE box visit (boxes differ by +-15 % in clocks)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fid"
{
for r in 1 2; do
for v in base cur; do
  unset CGAMD_LIB_PATH
  if [ $v != cur ]; then export CGAMD_LIB_PATH=compare_gan_amd/lib/libcgamd_$v.so; fi
  echo "$v $r: "; timeout 200 $B 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'])
for k,v in d['roofline']['kernels'].items(): print('    %-36s %7.3f ms  %6.1f us x %d' % (k, v['ms_per_step'], v['avg_launch_us'], v['launches_per_step']))
"
done; done
} > gpurun_out/ab.txt 2>&1
cat gpurun_out/ab.txt
