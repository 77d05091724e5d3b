cd $GRAFT_REPO_ROOT
C=L4_res0_raw,L4_res1_raw,L4_dec_res0_raw,L4_qkv_raw,L4_v_raw,L4_dec_skip_raw,L3_res0_raw,L3_res1_raw,L3_dec_res0_raw,L3_qkv_raw,L3_v_raw,L3_skip_cat_raw,L2_res0_raw,L2_skip_cat_raw
echo "== hot"; timeout 600 python tools/conv_bench.py --iters 80 --cases $C 2>&1 | grep -v amdgpu
echo "== cold (80 weight buffers)"; timeout 600 python tools/conv_bench.py --iters 80 --cold 80 --cases $C 2>&1 | grep -v amdgpu
