# round 6, first GPU call: tail kernels, double slab reduction, ATen-free helpers; A/B experiments
cd $GRAFT_REPO_ROOT
T=r6_a
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-multi-anchor --no-throughput-mode"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention or planes or split or conv or wgrad" 2>&1 | tail -8 > gpurun_out/${T}_pytest_ops.log
timeout 600 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "fp64" 2>&1 | grep -v "^$" | tail -12 > gpurun_out/${T}_fp64.log
timeout 600 $B > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
SVL_GEMM_EMU_H2_DENSE=1 timeout 600 $B --no-profile > gpurun_out/${T}_bench_h2dense.json 2>/dev/null
timeout 600 $B --no-profile > gpurun_out/${T}_bench_again.json 2>/dev/null
SVL_WGRAD_STREAM=1 timeout 600 $B --no-profile --as-multi > gpurun_out/${T}_bench_asmulti_wg.json 2>/dev/null
timeout 600 $B --no-profile --as-multi > gpurun_out/${T}_bench_asmulti.json 2>/dev/null
SVL_GEMM_EMU_H2_DENSE=1 timeout 600 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "fp64" 2>&1 | grep -v "^$" | tail -12 > gpurun_out/${T}_fp64_h2dense.log
TAG=$T timeout 900 bash tools/prof_step.sh > /dev/null 2>&1
for f in bench bench_h2dense bench_again bench_asmulti_wg bench_asmulti; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
tail -3 gpurun_out/${T}_pytest_ops.log; tail -4 gpurun_out/${T}_fp64.log | cut -c1-1500
