cd $GRAFT_REPO_ROOT
T=r6_e
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-multi-anchor --no-throughput-mode"
timeout 1800 python -m pytest tests -m gpu -q --durations=4 -s 2>&1 | grep -v "^$" | grep -E "passed|failed|FAILED|Error|RATCHET|BF16X3|RESULT|shared queue|queues shared|fraction|assert|ATTN_GATE" | cut -c1-500 | tail -70 > gpurun_out/${T}_pytest.log
timeout 600 $B > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
timeout 600 $B --no-profile > gpurun_out/${T}_bench_again.json 2>/dev/null
timeout 900 python bench.py --config ade --steps 3 --warmup 1 --no-cpu-baseline --no-multi-anchor --no-throughput-mode --no-profile > gpurun_out/${T}_bench_ade.json 2>/dev/null
timeout 900 python bench.py --config coco --steps 3 --warmup 1 --no-cpu-baseline --no-multi-anchor --no-throughput-mode --no-profile > gpurun_out/${T}_bench_coco.json 2>/dev/null
TAG=$T timeout 900 bash tools/prof_step.sh > /dev/null 2>&1
for f in bench bench_again bench_ade bench_coco; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
grep -v ATTN_GATE gpurun_out/${T}_pytest.log | cut -c1-500
