cd $GRAFT_REPO_ROOT
timeout 300 python tools/g1_check.py 2>&1 | grep -v amdgpu | tail -3
C=L3_v_raw,L3_proj_raw,L3_skip_cat_raw,L3_up_skip_raw,L3_qkv,L2_skip_cat_raw,L4_v_raw,L4_dec_skip_raw,L4_qkv
for cfg in "0 0" "2 2" "2 1" "1 2" "1 1"; do set -- $cfg; echo "== NST=8 MF=$1 NF=$2"; DDX_G1_MF=$1 DDX_G1_NF=$2 timeout 300 python tools/conv_bench.py --path g1 --cases $C 2>&1 | grep -v "amdgpu\|unsupp"; done
