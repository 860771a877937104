cd $GRAFT_REPO_ROOT
run2() { python bench.py "$@" --no-cpu-baseline --no-throughput-mode --no-profile --no-multi-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['peak_mem_gb'], d['config']['loss'])"; }
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "planes or layernorm or emulation" 2>&1 | tail -4
for nf in 1 0 1 0; do echo -n "VOC h2 nofast=$nf: "; SVL_PLANES_NO_FAST_EPI=$nf run2 --steps 8 --warmup 3; done
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -s 2>&1 | grep -E "gemm_mode|passed|failed|Error" | cut -c1-1800 | tail -20
