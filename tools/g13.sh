cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "fused_attention or class_sequences" 2>&1 | tail -3
bash tools/g12.sh
