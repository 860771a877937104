cd $GRAFT_REPO_ROOT
run2() { python bench.py "$@" --no-cpu-baseline --no-throughput-mode --no-profile --no-multi-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['peak_mem_gb'], d['config']['loss'])"; }
for f in h2 b3 h2 b3; do echo -n "VOC attn $f: "; SVL_ATTN_FMT=$f run2 --steps 8 --warmup 3; done
for f in h2 b3; do echo -n "cityscapes attn $f: "; SVL_ATTN_FMT=$f run2 --config cityscapes --steps 4 --warmup 2; done
TAG=r5_c bash tools/prof_step.sh > /dev/null 2>&1
head -30 gpurun_out/r5_c_kernel_stats.csv | cut -c1-150
