# kernel-trace summary of the bench step (2 steps + 1 warm-up, no per-kernel event profiling): usage  TAG=r3_a bash tools/prof_step.sh [bench args]
T=${TAG:-r3}
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$T -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-multi-anchor --no-profile "$@" > $GRAFT_REPO_ROOT/gpurun_out/${T}_bench_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT; DB=$(find gpurun_out/prof_$T -name "*.db" | head -1)
python tools/rocpd_stats.py $DB > gpurun_out/${T}_kernel_stats.csv
python tools/rocpd_attrib.py $DB 0.34 1.0 > gpurun_out/${T}_attrib.txt
python tools/rocpd_timeline.py $DB 0.34 1.0 >> gpurun_out/${T}_attrib.txt
rm -rf gpurun_out/prof_$T
cat gpurun_out/${T}_attrib.txt
