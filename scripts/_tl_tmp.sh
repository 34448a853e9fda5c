cd $GRAFT_REPO_ROOT
KSEL="(test_gconv_forward_adjoint_wgrad and (pc_ or fast_big or hc_)) or test_gconv_gates_residual or (test_conv_pool_fused and not full_size) or (test_gconv_fused_batch_norm and not full_size) or test_gconv_fused_statistics_groups"
CGAMD_QCONV=1 CGAMD_QCONV_MIN=1 CGAMD_HCONV_MIN=1 CGAMD_HCONV_RW=0 timeout 900 python -m pytest -q -m gpu tests/test_kernels_gpu.py -x -k "$KSEL" 2>&1 | tail -5
CGAMD_QCONV=1 timeout 900 python -m pytest -q -m gpu tests/test_kernels_gpu.py -x -k "full_size" 2>&1 | tail -4
bash scripts/gpu.sh pconv_ab ab4 CGAMD_QCONV=1 CGAMD_PCONV=1 CGAMD_QCONV=0
