# kernel-trace summary of the bench step (2 steps + 1 warm-up): usage  TAG=r3_a bash tools/prof_step.sh [bench args]
T=${TAG:-r3}
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$T -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode "$@" > $GRAFT_REPO_ROOT/gpurun_out/${T}_bench_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT; python tools/rocpd_stats.py $(find gpurun_out/prof_$T -name "*.db" | head -1) > gpurun_out/${T}_kernel_stats.csv; rm -rf gpurun_out/prof_$T
head -40 gpurun_out/${T}_kernel_stats.csv | cut -c1-150
