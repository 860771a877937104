cd $GRAFT_REPO_ROOT
T=r6_c
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-multi-anchor --no-throughput-mode"
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_multiproc_gpu.py -q -s -k "convT or conv_t or transpose or up_block or collective or shortk or short" 2>&1 | grep -E "passed|failed|FAILED|RESULT|fraction|queues shared|assert" | cut -c1-400 | tail -12 > gpurun_out/${T}_pytest_a.log
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_model_gpu.py -q -s -x 2>&1 | grep -E "passed|failed|FAILED|FP64 RATCHET|BF16X3|assert" | cut -c1-700 | tail -12 > gpurun_out/${T}_pytest_b.log
timeout 600 $B > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
SVL_SHORTK_NO_GATHER=1 timeout 600 $B --no-profile > gpurun_out/${T}_bench_nogather.json 2>/dev/null
SVL_GEMM_EMU_H2_DENSE=0 timeout 600 $B --no-profile > gpurun_out/${T}_bench_nodense.json 2>/dev/null
timeout 600 $B --no-profile > gpurun_out/${T}_bench_again.json 2>/dev/null
timeout 900 python bench.py --config ade --steps 3 --warmup 1 --no-cpu-baseline --no-multi-anchor --no-throughput-mode --no-profile > gpurun_out/${T}_bench_ade.json 2>/dev/null
SVL_SHORTK_NO_GATHER=1 timeout 900 python bench.py --config ade --steps 3 --warmup 1 --no-cpu-baseline --no-multi-anchor --no-throughput-mode --no-profile > gpurun_out/${T}_bench_ade_nogather.json 2>/dev/null
TAG=$T timeout 900 bash tools/prof_step.sh > /dev/null 2>&1
for f in bench bench_nogather bench_nodense bench_again bench_ade bench_ade_nogather; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
cat gpurun_out/${T}_pytest_a.log gpurun_out/${T}_pytest_b.log | cut -c1-700
grep "shortk\|gemm_kernelILi128ELi64\|gemm_kernelILi64ELi128" gpurun_out/${T}_kernel_stats.csv | cut -c1-150
