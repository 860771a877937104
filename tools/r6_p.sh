cd $GRAFT_REPO_ROOT
T=r6_p
timeout 900 python -m pytest tests/test_ops_gpu.py -q -s -k "tiled_wgrad_running or tiled_narrow or consuming_conv or conv3x3" 2>&1 | grep -E "passed|failed|FAILED|WGRAD|assert|Error|error" | cut -c1-300 | tail -30 > gpurun_out/${T}_pytest_a.log
cat gpurun_out/${T}_pytest_a.log
