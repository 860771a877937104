cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_attn -o attn -- python $GRAFT_REPO_ROOT/tools/bench_attn.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; DB=$(find gpurun_out/prof_attn -name "*.db" | head -1)
python tools/rocpd_stats.py $DB | head -16 | cut -c1-160
rm -rf gpurun_out/prof_attn
