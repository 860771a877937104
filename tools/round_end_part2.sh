# second half of the evidence run (per-config lines, exact-f32 line, kernel summaries); stderr kept
cd $GRAFT_REPO_ROOT
T=${TAG:-r3}
for c in cityscapes ade coco; do python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-mode > gpurun_out/${T}_bench_$c.json 2> gpurun_out/${T}_bench_$c.err; done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-throughput-mode --gemm-arith f32 > gpurun_out/${T}_bench_exact_f32.json 2> gpurun_out/${T}_bench_exact_f32.err
for m in bf16x6 f32; do
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$m -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-profile --gemm-arith $m > $GRAFT_REPO_ROOT/gpurun_out/${T}_bench_under_rocprof_$m.json 2> $GRAFT_REPO_ROOT/gpurun_out/${T}_rocprof_$m.err
  cd $GRAFT_REPO_ROOT; DB=$(find gpurun_out/prof_$m -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB > gpurun_out/${T}_kernel_stats_bs16_$m.csv
  python tools/rocpd_attrib.py $DB 0.34 1.0 > gpurun_out/${T}_attrib_$m.txt
  if [ $m = bf16x6 ]; then python tools/rocpd_dispatches.py $DB gemm_x6p_kernelILi256ELi1E 1548 > gpurun_out/${T}_dominant_dispatches.csv; fi
  rm -rf gpurun_out/prof_$m
done
for f in cityscapes ade coco exact_f32; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_$f.json').read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$f', d['value'], d['ms_per_step'], d['config']['peak_mem_gb'], r.get('frac'), r.get('avg_ms'))"; done
tail -3 gpurun_out/${T}_bench_*.err | tail -20
