cd $GRAFT_REPO_ROOT
T=r6_o
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-multi-anchor --no-throughput-mode --no-profile"
A="python bench.py --config ade --steps 3 --warmup 1 --no-cpu-baseline --no-multi-anchor --no-throughput-mode --no-profile"
timeout 900 python -m pytest tests/test_ops_gpu.py -q -s -k "wgrad or consuming_conv or conv_fwd_dgrad" 2>&1 | grep -E "passed|failed|FAILED|WGRAD|assert|Error|error" | cut -c1-300 | tail -30 > gpurun_out/${T}_pytest_a.log
cat gpurun_out/${T}_pytest_a.log
timeout 600 $B > gpurun_out/${T}_bench.json 2>/dev/null
SVL_CONV_WGRAD_NO_H2=1 timeout 600 $B > gpurun_out/${T}_bench_nowgh2.json 2>/dev/null
timeout 600 $B > gpurun_out/${T}_bench_again.json 2>/dev/null
timeout 900 $A > gpurun_out/${T}_bench_ade.json 2>/dev/null
SVL_CONV_WGRAD_NO_H2=1 timeout 900 $A > gpurun_out/${T}_bench_ade_nowgh2.json 2>/dev/null
for f in bench bench_nowgh2 bench_again bench_ade bench_ade_nowgh2; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
