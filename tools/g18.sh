cd $GRAFT_REPO_ROOT
run2() { python bench.py "$@" --no-cpu-baseline --no-throughput-mode --no-profile --no-multi-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['peak_mem_gb'], d['config']['loss'])"; }
for c in ade coco; do for f in h2 b3; do echo -n "$c attn $f: "; SVL_ATTN_FMT=$f run2 --config $c --steps 3 --warmup 1; done; done
timeout 2300 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r5_d_pytest.log
