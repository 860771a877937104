cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ops_gpu.py -q -m gpu -x 2>&1 | tail -6
run2() { python bench.py "$@" --no-cpu-baseline --no-throughput-mode --no-profile --no-multi-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['peak_mem_gb'], d['config']['loss'])"; }
for nh in 0 1 0 1; do echo -n "VOC no_h2=$nh: "; if [ $nh = 1 ]; then SVL_GEMM_EMU_NO_H2=1 run2 --steps 8 --warmup 3; else run2 --steps 8 --warmup 3; fi; done
for nh in 0 1; do echo -n "ADE no_h2=$nh: "; if [ $nh = 1 ]; then SVL_GEMM_EMU_NO_H2=1 run2 --config ade --steps 3 --warmup 1; else run2 --config ade --steps 3 --warmup 1; fi; done
