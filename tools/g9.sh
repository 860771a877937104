cd $GRAFT_REPO_ROOT
for f in h2 b3; do echo "== SVL_ATTN_FMT=$f"; SVL_ATTN_FMT=$f timeout 300 python tools/bench_attn.py 2>&1 | tail -3; done
SVL_ATTN_FMT=h2 timeout 300 python tools/bench_attn.py 2 2602 12 2>&1 | tail -2
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fused_attention or class_sequences" 2>&1 | tail -15
