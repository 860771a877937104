cd $GRAFT_REPO_ROOT
T=r6_j
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-multi-anchor --no-throughput-mode --no-profile"
A="python bench.py --config ade --steps 3 --warmup 1 --no-cpu-baseline --no-multi-anchor --no-throughput-mode --no-profile"
timeout 900 python -m pytest tests/test_ops_gpu.py -q -s -k "conv or groupnorm or gn" 2>&1 | grep -E "passed|failed|FAILED|TILED_H2|assert|Error" | cut -c1-300 | tail -25 > gpurun_out/${T}_pytest_a.log
cat gpurun_out/${T}_pytest_a.log
timeout 600 $B > gpurun_out/${T}_bench.json 2>/dev/null
SVL_CONV_TILED_NO_H2=1 timeout 600 $B > gpurun_out/${T}_bench_noh2.json 2>/dev/null
timeout 600 $B > gpurun_out/${T}_bench_again.json 2>/dev/null
timeout 900 $A > gpurun_out/${T}_bench_ade.json 2>/dev/null
SVL_CONV_TILED_NO_H2=1 timeout 900 $A > gpurun_out/${T}_bench_ade_noh2.json 2>/dev/null
for f in bench bench_noh2 bench_again bench_ade bench_ade_noh2; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -q -s 2>&1 | grep -E "passed|failed|FAILED|RATCHET|assert" | cut -c1-400 | tail -25
TAG=$T timeout 900 bash tools/prof_step.sh > /dev/null 2>&1
grep "conv3x3_tiled" gpurun_out/${T}_kernel_stats.csv | cut -c1-150
