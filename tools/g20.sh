cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "memory_guard" 2>&1 | grep -E "^E|passed|failed" | head -20
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "emits_planes" 2>&1 | grep -E "^E|passed|failed" | head -20
run2() { python bench.py "$@" --no-cpu-baseline --no-throughput-mode --no-profile --no-multi-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['peak_mem_gb'], d['config']['loss'])"; }
for rp in 0 1 0 1; do echo -n "VOC norepack=$rp: "; if [ $rp = 1 ]; then SVL_ATTN_NO_REPACK=1 run2 --steps 8 --warmup 3; else run2 --steps 8 --warmup 3; fi; done
