# End-of-round evidence run on the 1-GPU box (profiles/README.md):  gpurun -- 'TAG=r5_x bash tools/round_end_run.sh'
cd $GRAFT_REPO_ROOT
T=${TAG:-r5}
python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -25 > gpurun_out/${T}_pytest.log
# PMC records of the dominant launches (quoted by bench.py as roofline.traffic / roofline_hbm.traffic)
X6P_FMT=1 bash tools/pmc_x6p.sh 32800 3072 768 4 > gpurun_out/${T}_pmc_x6p_ffn1.txt 2>&1   # mode 4 = FFN-1 as the step launches it (no fp32 C); fmt 1 = fp16 x 2 planes
bash tools/pmc_ce.sh > gpurun_out/${T}_pmc_ce.txt 2>&1
bash tools/pmc_sq.sh tools/one_ce_up.py ce_up_kernel > gpurun_out/${T}_pmc_sq_ce_up.txt 2>&1   # SQ counters of the kernel the step launches for its pixel losses
SVL_GEMM_EMU=0 bash tools/pmc_traffic.sh tools/one_gemm.py gemm_kernel > gpurun_out/${T}_pmc_gemm_f32.txt 2>&1
mkdir -p profiles && cp gpurun_out/pmc_x6p_traffic.json gpurun_out/pmc_ce_traffic.json gpurun_out/pmc_gemm_traffic.json profiles/ 2>/dev/null
# kernel-trace summaries (both arithmetics) + wall-time attribution + per-dispatch rows of the dominant kernel
for m in bf16x6 f32; do
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$m -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-profile --gemm-arith $m > $GRAFT_REPO_ROOT/gpurun_out/${T}_bench_under_rocprof_$m.json 2>/dev/null
  cd $GRAFT_REPO_ROOT; DB=$(find gpurun_out/prof_$m -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB > gpurun_out/${T}_kernel_stats_bs16_$m.csv
  python tools/rocpd_attrib.py $DB 0.34 1.0 > gpurun_out/${T}_attrib_$m.txt
  if [ $m = bf16x6 ]; then
    python tools/rocpd_dispatches.py $DB gemm_x6p_kernelILi2ELi256ELi1E 1548 > gpurun_out/${T}_dominant_dispatches.csv
    python tools/dispatch_record.py gpurun_out/${T}_dominant_dispatches.csv ${T} > gpurun_out/dominant_dispatches.json && cp gpurun_out/dominant_dispatches.json profiles/
  fi
  rm -rf gpurun_out/prof_$m
done
python bench.py --steps 10 --warmup 3 > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
for c in cityscapes ade coco; do python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-mode > gpurun_out/${T}_bench_$c.json 2>/dev/null; done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-throughput-mode --gemm-arith f32 > gpurun_out/${T}_bench_exact_f32.json 2>/dev/null
# the N > 1 code path on this one-GPU box: two ranks over gloo (blocking all-reduce) -- the `allreduce` object with the
# stream -> hardware-queue probe of the reducer
SVL_DIST_BACKEND=gloo python bench.py --gpus 2 --batch 4 --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-profile > gpurun_out/${T}_bench_gloo2.json 2> gpurun_out/${T}_bench_gloo2.err
# every MFMA launch shape of the VOC and ADE steps (solo HIP-event durations)
python tools/shape_table.py voc > gpurun_out/${T}_shapes_voc.txt 2>/dev/null
python tools/shape_table.py ade > gpurun_out/${T}_shapes_ade.txt 2>/dev/null
for c in ade coco cityscapes; do TAG=${T}_$c bash tools/prof_step.sh --config $c > /dev/null 2>&1; done
tail -4 gpurun_out/${T}_pytest.log
for f in default cityscapes ade coco exact_f32; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_$f.json').read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$f', d['value'], d['ms_per_step'], d['config']['peak_mem_gb'], r.get('frac'), r.get('avg_ms'), (d.get('exact_f32') or {}).get('value'))"; done
