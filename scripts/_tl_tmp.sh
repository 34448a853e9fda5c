cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest -q -m gpu tests/test_kernels_gpu.py -x -k "test_gconv_forward_adjoint_wgrad or test_gconv_gates_residual or full_size or test_conv_pool_fused or test_stem_relu_gate" 2>&1 | tail -4
for L in prev new prev new; do
  echo "== $L"
  if [ $L = prev ]; then export CGAMD_LIB_PATH=$GRAFT_REPO_ROOT/compare_gan_amd/lib/libcgamd_prev.so; else unset CGAMD_LIB_PATH; fi
  timeout 600 python scripts/bench_convs.py hc 2>&1 | grep -v amdgpu | head -8
done
