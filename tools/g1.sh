cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "planes or layernorm" 2>&1 | tail -30
