cd $GRAFT_REPO_ROOT
T=r05_v11
timeout 900 python -m pytest -q -m gpu tests/test_modular_gan_gpu.py -x -k "train_steps_resnet_cifar or joint_gen_for_disc_step or not_unrolled or c5_batch" --durations=8 2>&1 | tail -25 > gpurun_out/${T}_t_gan.txt
timeout 300 python -m pytest -q -m gpu tests/test_eval_gpu.py -x -k "nan" 2>&1 | tail -5 > gpurun_out/${T}_t_eval.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3 > gpurun_out/${T}_smoke.txt
for f in gpurun_out/${T}_t_gan.txt gpurun_out/${T}_t_eval.txt gpurun_out/${T}_smoke.txt; do echo "== $f"; tail -n 14 $f; done
