cd $GRAFT_REPO_ROOT
t() { echo "== $1"; env $1 timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -s -k "gradient_error_against_fp64" 2>&1 | grep -E "worst ratios|^E  .*mode|passed|failed" | cut -c1-700; }
t X=1
t SVL_GEMM_EMU_NO_H2=1
t SVL_LN_NO_FUSED_H2=1
t SVL_ATTN_NO_REPACK=1
t SVL_ATTN_FMT=b3
