cd $GRAFT_REPO_ROOT
T=r6_d
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-multi-anchor --no-throughput-mode"
A="python bench.py --config ade --steps 3 --warmup 1 --no-cpu-baseline --no-multi-anchor --no-throughput-mode --no-profile"
timeout 1500 python -m pytest tests -m gpu -q --durations=4 -s 2>&1 | grep -v "^$" | grep -E "passed|failed|FAILED|Error|RATCHET|BF16X3|RESULT|shared queue|queues shared|fraction|assert" | cut -c1-500 | tail -50 > gpurun_out/${T}_pytest.log
timeout 600 $B > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
SVL_NO_GN_BWD_FUSED=1 timeout 600 $B --no-profile > gpurun_out/${T}_bench_nognb.json 2>/dev/null
SVL_GEMM_EMU_H2_CONVFWD=1 timeout 600 $B --no-profile > gpurun_out/${T}_bench_convfwd.json 2>/dev/null
timeout 600 $B --no-profile > gpurun_out/${T}_bench_again.json 2>/dev/null
timeout 900 $A > gpurun_out/${T}_bench_ade.json 2>/dev/null
SVL_NO_GN_BWD_FUSED=1 timeout 900 $A > gpurun_out/${T}_bench_ade_nognb.json 2>/dev/null
SVL_GEMM_EMU_H2_CONVFWD=1 timeout 900 $A > gpurun_out/${T}_bench_ade_convfwd.json 2>/dev/null
SVL_GEMM_EMU_H2_CONVFWD=1 timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -s -k "fp64 or matches_oracle" 2>&1 | grep -E "passed|failed|FAILED|FP64 RATCHET|RATCHET measured|assert" | cut -c1-600 > gpurun_out/${T}_convfwd_tests.log
cp gpurun_out/numerics_fp64.json gpurun_out/${T}_numerics_convfwd.json
TAG=$T timeout 900 bash tools/prof_step.sh > /dev/null 2>&1
for f in bench bench_nognb bench_convfwd bench_again bench_ade bench_ade_nognb bench_ade_convfwd; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
cat gpurun_out/${T}_pytest.log | cut -c1-500; echo ---; cat gpurun_out/${T}_convfwd_tests.log
