cd $GRAFT_REPO_ROOT
T=r6_n
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -q -s 2>&1 | grep -E "passed|failed|FAILED|RATCHET|ratchet|assert|Error|fp64|FP64" | cut -c1-500 | tail -40 > gpurun_out/${T}_pytest_b.log
cat gpurun_out/${T}_pytest_b.log
cp profiles/numerics_fp64.json gpurun_out/${T}_numerics_fp64.json 2>/dev/null
