cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -25 > gpurun_out/r2e_pytest.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r2e_bench_default.json 2> gpurun_out/r2e_bench_default.err
for c in cityscapes ade coco; do python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2e_bench_$c.json 2>/dev/null; done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-throughput-mode --gemm-arith bf16x6 > gpurun_out/r2e_bench_bf16x6.json 2>/dev/null
SVL_GEMM_EMU=0 bash tools/pmc_traffic.sh tools/one_gemm.py gemm_kernel > gpurun_out/r2e_pmc_traffic.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2e -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode > $GRAFT_REPO_ROOT/gpurun_out/r2e_bench_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT; python tools/rocpd_stats.py $(find gpurun_out/prof_r2e -name "*.db" | head -1) > gpurun_out/r2e_kernel_stats_bs16.csv; rm -rf gpurun_out/prof_r2e
tail -12 gpurun_out/r2e_pytest.log; cat gpurun_out/r2e_pmc_traffic.txt | tail -4
for f in default cityscapes ade coco bf16x6; do python -c "
import json; d=json.load(open('gpurun_out/r2e_bench_$f.json')); print('$f', d['value'], d['ms_per_step'], d['config']['peak_mem_gb'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('frac_whole_step'))"; done
