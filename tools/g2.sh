cd $GRAFT_REPO_ROOT
. tools/ab_lib.sh
run2() { python bench.py "$@" --no-cpu-baseline --no-throughput-mode --no-profile --no-multi-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['peak_mem_gb'], d['config']['loss'])"; }
for f in b3 h2 b3 h2; do echo -n "VOC fmt $f: "; SVL_PLANES_FMT=$f run2 --steps 8 --warmup 3; done
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
