# End-of-round evidence run on the 1-GPU box:  gpurun -- 'TAG=r6_x bash tools/round_end_run.sh'   (everything lands in gpurun_out/)
cd $GRAFT_REPO_ROOT
T=${TAG:-r6}
python -m pytest tests -m gpu -q --durations=8 -s 2>&1 | grep -v "^$" | grep -E "passed|failed|FAILED|RATCHET|BF16X3|RESULT|s call|s setup" | cut -c1-400 | tail -40 > gpurun_out/${T}_pytest.log
cp gpurun_out/numerics_fp64.json gpurun_out/${T}_numerics_fp64.json 2>/dev/null
# the C-ABI harness of the planes GEMM (git-ignored binary: built here)
[ -x tools/micro/x6p_bench ] || hipcc --offload-arch=gfx950 -O2 tools/micro/x6p_bench.cpp -Iinclude -Lsemivl_amd -lsemivl_hip -Wl,-rpath,$GRAFT_REPO_ROOT/semivl_amd -o tools/micro/x6p_bench > gpurun_out/${T}_x6p_build.log 2>&1
# PMC records (separate --pmc passes): the dominant launch's counters + traffic, the in-step pixel-loss kernel's SQ record, the
# three fused-attention grids' SQ record (SQ_VALU_MFMA_BUSY_CYCLES, SQ_WAIT_INST_LDS, GRBM_GUI_ACTIVE ...), the exact fp32 GEMM
X6P_FMT=1 bash tools/pmc_x6p.sh 32800 3072 768 4 > gpurun_out/${T}_pmc_x6p_ffn1.txt 2>&1
bash tools/pmc_sq.sh tools/one_ce_up.py ce_up_kernel > gpurun_out/${T}_pmc_sq_ce_up.txt 2>&1
for k in attn_fwd_h2_kernel attn_dkv_h2_kernel attn_dq_h2_kernel; do EMU=6 bash tools/pmc_sq.sh tools/one_attn.py $k; done > gpurun_out/${T}_pmc_sq_attn_h2.txt 2>&1
SVL_GEMM_EMU=0 bash tools/pmc_traffic.sh tools/one_gemm.py gemm_kernel > gpurun_out/${T}_pmc_gemm_f32.txt 2>&1
# kernel-trace summaries (both arithmetics) + wall-time attribution + per-dispatch rows of the dominant kernel + attention means
for m in bf16x6 f32; do
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$m -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-multi-anchor --no-profile --gemm-arith $m > $GRAFT_REPO_ROOT/gpurun_out/${T}_bench_under_rocprof_$m.json 2>/dev/null
  cd $GRAFT_REPO_ROOT; DB=$(find gpurun_out/prof_$m -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB > gpurun_out/${T}_kernel_stats_bs16_$m.csv
  python tools/rocpd_attrib.py $DB 0.34 1.0 > gpurun_out/${T}_attrib_$m.txt
  if [ $m = bf16x6 ]; then
    python tools/rocpd_dispatches.py $DB gemm_x6p_kernelILi2ELi256ELi1E 1548 > gpurun_out/${T}_dominant_dispatches.csv
    python tools/dispatch_record.py gpurun_out/${T}_dominant_dispatches.csv ${T} > gpurun_out/dominant_dispatches.json
    python tools/inloop_record.py $DB ${T} > gpurun_out/inloop_dispatches.json
  fi
  rm -rf gpurun_out/prof_$m
done
# the records bench.py quotes must be in profiles/ BEFORE the default line is taken
mkdir -p profiles && cp gpurun_out/pmc_x6p_traffic.json gpurun_out/pmc_gemm_traffic.json gpurun_out/dominant_dispatches.json gpurun_out/inloop_dispatches.json gpurun_out/numerics_fp64.json profiles/ 2>/dev/null
python bench.py --steps 10 --warmup 3 > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
for c in cityscapes ade coco; do python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-multi-anchor > gpurun_out/${T}_bench_$c.json 2>/dev/null; done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-throughput-mode --no-multi-anchor --gemm-arith f32 > gpurun_out/${T}_bench_exact_f32.json 2>/dev/null
# BASELINE configs[4] names "fp16 mixed precision": a build-only mode, no reference counterpart -- recorded as such
python bench.py --config coco --gemm-arith bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-multi-anchor > gpurun_out/${T}_bench_coco_bf16x3_build_only_mode.json 2>/dev/null
# the N > 1 code path on this one-GPU box: two ranks over gloo (blocking all-reduce), and the stream-ordered branch traced
SVL_DIST_BACKEND=gloo python bench.py --gpus 2 --batch 4 --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-profile > gpurun_out/${T}_bench_gloo2.json 2> gpurun_out/${T}_bench_gloo2.err
TAG=$T SVL_COMM_B=16 bash tools/comm_overlap_trace.sh > /dev/null 2>&1
python tools/shape_table.py voc > gpurun_out/${T}_shapes_voc.txt 2>/dev/null
python tools/shape_table.py ade > gpurun_out/${T}_shapes_ade.txt 2>/dev/null
for c in ade coco cityscapes; do TAG=${T}_$c bash tools/prof_step.sh --config $c > /dev/null 2>&1; done
tail -4 gpurun_out/${T}_pytest.log
for f in default cityscapes ade coco exact_f32 coco_bf16x3_build_only_mode; do python -c "
import json; d=json.loads(open('gpurun_out/${T}_bench_$f.json').read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$f', d['value'], d['ms_per_step'], d['config']['peak_mem_gb'], r.get('frac'), r.get('avg_ms'), (d.get('exact_f32') or {}).get('value'))"; done
tail -3 gpurun_out/${T}_comm_overlap.txt
