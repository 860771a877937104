# End-of-round evidence run on the 1-GPU box (profiles/README.md, `r2_g_*`):  gpurun -- 'bash tools/round_end_run.sh'
cd $GRAFT_REPO_ROOT
T=${TAG:-r2g}
python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -25 > gpurun_out/${T}_pytest.log
SVL_GEMM_EMU=0 bash tools/pmc_traffic.sh tools/one_gemm.py gemm_kernel > gpurun_out/${T}_pmc_traffic.txt 2>&1
SVL_GEMM_EMU=6 bash tools/pmc_traffic.sh tools/one_gemm.py gemm_bf16x >> gpurun_out/${T}_pmc_traffic.txt 2>&1
mkdir -p profiles && cp gpurun_out/pmc_gemm_traffic.json gpurun_out/pmc_gemm_traffic_bf16x6.json profiles/   # (so that the bench lines below can quote them)
python bench.py --steps 10 --warmup 3 > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
for c in cityscapes ade coco; do python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_bench_$c.json 2>/dev/null; done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-throughput-mode --gemm-arith f32 > gpurun_out/${T}_bench_exact_f32.json 2>/dev/null
for m in bf16x6 f32; do
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$m -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode --gemm-arith $m > $GRAFT_REPO_ROOT/gpurun_out/${T}_bench_under_rocprof_$m.json 2>/dev/null
  cd $GRAFT_REPO_ROOT; python tools/rocpd_stats.py $(find gpurun_out/prof_$m -name "*.db" | head -1) > gpurun_out/${T}_kernel_stats_bs16_$m.csv; rm -rf gpurun_out/prof_$m
done
tail -6 gpurun_out/${T}_pytest.log; tail -3 gpurun_out/${T}_pmc_traffic.txt
for f in default cityscapes ade coco exact_f32; do python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_$f.json')); r=d.get('roofline',{}); print('$f', d['value'], d['ms_per_step'], d['config']['peak_mem_gb'], r.get('frac'), r.get('frac_whole_step'), (d.get('exact_f32') or {}).get('value'))"; done
