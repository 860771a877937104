cd $GRAFT_REPO_ROOT
T=r6_g
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-multi-anchor --no-throughput-mode --no-profile"
A="python bench.py --config ade --steps 3 --warmup 1 --no-cpu-baseline --no-multi-anchor --no-throughput-mode --no-profile"
timeout 600 python -m pytest tests/test_ops_gpu.py -q -s -k "convT or conv_t" 2>&1 | grep -E "passed|failed|FAILED|CONVT_DGRAD|assert|Error" | cut -c1-400 | tail -12 > gpurun_out/${T}_pytest_a.log
cat gpurun_out/${T}_pytest_a.log
timeout 600 $B > gpurun_out/${T}_bench.json 2>/dev/null
SVL_CONVT_NO_TILED=1 timeout 600 $B > gpurun_out/${T}_bench_noct.json 2>/dev/null
timeout 600 $B > gpurun_out/${T}_bench_again.json 2>/dev/null
timeout 900 $A > gpurun_out/${T}_bench_ade.json 2>/dev/null
SVL_CONVT_NO_TILED=1 timeout 900 $A > gpurun_out/${T}_bench_ade_noct.json 2>/dev/null
TAG=$T timeout 900 bash tools/prof_step.sh > /dev/null 2>&1
for f in bench bench_noct bench_again bench_ade bench_ade_noct; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
grep "convt2x\|gemm_kernelILi128ELi64ELi2ELi2ELi2\|gemm_kernelILi64ELi128" gpurun_out/${T}_kernel_stats.csv | cut -c1-150
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -3
