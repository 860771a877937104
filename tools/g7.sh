cd $GRAFT_REPO_ROOT
B=tools/micro/x6p_phases
for mode in 0 4; do for np in 2 3; do timeout 120 $B 32768 3072 768 $mode $np; done; done
timeout 120 $B 32768 768 3072 0 2
timeout 120 $B 32768 768 768 0 2
timeout 120 $B 32768 2304 768 0 2
NO_FAST=1 timeout 120 $B 32768 3072 768 4 2
B=tools/micro/x6p_bench
for args in "32800 3072 768 20 0" "32800 3072 768 20 4" "32800 3072 768 20 3" "32800 768 768 20 2" "32800 768 3072 20 2" "32800 2304 768 20 0" "32800 768 2304 20 0"; do
  for f in 1 0; do echo "fmt $f: $(X6P_FMT=$f timeout 120 $B $args | tr '\n' ' ')"; done
done
run2() { python bench.py "$@" --no-cpu-baseline --no-throughput-mode --no-profile --no-multi-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['peak_mem_gb'], d['config']['loss'])"; }
echo -n "VOC: "; run2 --steps 8 --warmup 3
