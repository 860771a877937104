cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "layernorm or gemm_planes_path" 2>&1 | tail -3
run2() { python bench.py "$@" --no-cpu-baseline --no-throughput-mode --no-profile --no-multi-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['peak_mem_gb'], d['config']['loss'])"; }
for nf in 0 1 0 1; do echo -n "VOC ln_no_fused=$nf: "; if [ $nf = 1 ]; then SVL_LN_NO_FUSED_H2=1 run2 --steps 8 --warmup 3; else run2 --steps 8 --warmup 3; fi; done
