# round 6, second GPU call: full GPU suite; 4-wave tails; k2s2 gather A/B; as-multi weight-gradient stream A/B
cd $GRAFT_REPO_ROOT
T=r6_b
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-multi-anchor --no-throughput-mode"
timeout 1500 python -m pytest tests -m gpu -q --durations=6 -s 2>&1 | grep -v "^$" | grep -E "passed|failed|FAILED|Error|error|RATCHET|BF16X3|worst ratios|recomputed|RESULT|assert" | cut -c1-1800 | tail -60 > gpurun_out/${T}_pytest.log
timeout 600 $B > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
SVL_SHORTK_NO_GATHER=1 timeout 600 $B --no-profile > gpurun_out/${T}_bench_nogather.json 2>/dev/null
timeout 600 $B --no-profile > gpurun_out/${T}_bench_again.json 2>/dev/null
SVL_WGRAD_STREAM=1 timeout 600 $B --no-profile --as-multi > gpurun_out/${T}_bench_asmulti_wg.json 2>/dev/null
timeout 600 $B --no-profile --as-multi > gpurun_out/${T}_bench_asmulti.json 2>/dev/null
timeout 900 python bench.py --config ade --steps 3 --warmup 1 --no-cpu-baseline --no-multi-anchor --no-throughput-mode --no-profile > gpurun_out/${T}_bench_ade.json 2>/dev/null
SVL_SHORTK_NO_GATHER=1 timeout 900 python bench.py --config ade --steps 3 --warmup 1 --no-cpu-baseline --no-multi-anchor --no-throughput-mode --no-profile > gpurun_out/${T}_bench_ade_nogather.json 2>/dev/null
TAG=$T timeout 900 bash tools/prof_step.sh > /dev/null 2>&1
for f in bench bench_nogather bench_again bench_asmulti_wg bench_asmulti bench_ade bench_ade_nogather; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
cat gpurun_out/${T}_pytest.log | cut -c1-600
