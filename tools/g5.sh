cd $GRAFT_REPO_ROOT
TAG=r5_b bash tools/prof_step.sh > /dev/null 2>&1
head -45 gpurun_out/r5_b_kernel_stats.csv 2>/dev/null || ls gpurun_out | grep r5_b
cat gpurun_out/r5_b_attrib*.txt 2>/dev/null | head -40
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -s -k "batch_invariant or matches_oracle or large_class" 2>&1 | grep -E "RATCHET|passed|failed|Error|assert" | cut -c1-300 | tail -20
