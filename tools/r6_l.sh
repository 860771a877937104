cd $GRAFT_REPO_ROOT
T=r6_l
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-multi-anchor --no-throughput-mode --no-profile"
A="python bench.py --config ade --steps 3 --warmup 1 --no-cpu-baseline --no-multi-anchor --no-throughput-mode --no-profile"
timeout 900 python -m pytest tests/test_ops_gpu.py -q -s -x -k "dilated or shortk or short_k or convT or conv_fwd_dgrad or upsample_convT" 2>&1 | grep -E "passed|failed|FAILED|DIL_H2|assert|Error|error" | cut -c1-300 | tail -40 > gpurun_out/${T}_pytest_a.log
cat gpurun_out/${T}_pytest_a.log
timeout 600 $B > gpurun_out/${T}_bench.json 2>/dev/null
SVL_CONV_NO_DIL=1 timeout 600 $B > gpurun_out/${T}_bench_nodil.json 2>/dev/null
SVL_SHORTK_NO_H2=1 timeout 600 $B > gpurun_out/${T}_bench_noskh2.json 2>/dev/null
timeout 900 $A > gpurun_out/${T}_bench_ade.json 2>/dev/null
SVL_CONV_NO_DIL=1 timeout 900 $A > gpurun_out/${T}_bench_ade_nodil.json 2>/dev/null
SVL_SHORTK_NO_H2=1 timeout 900 $A > gpurun_out/${T}_bench_ade_noskh2.json 2>/dev/null
for f in bench bench_nodil bench_noskh2 bench_ade bench_ade_nodil bench_ade_noskh2; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
