cd $GRAFT_REPO_ROOT
T=r05_v13
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_tl && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --preheat-s 0 --no-cpu-baseline --no-roofline --no-fid --no-legs > $GRAFT_REPO_ROOT/gpurun_out/${T}_tl.log 2>&1 )
F=$(find /tmp/p_tl -name '*kernel_trace.csv' | head -1)
wc -l $F
head -2 $F | cut -c1-400
python scripts/graph_timeline.py $F 480 | tee gpurun_out/${T}_cifar_graph_timeline.txt
tail -c 600 gpurun_out/${T}_tl.log
