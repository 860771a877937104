cd /tmp && export TMPDIR=/tmp
for v in 1 3 1 3; do
  SVL_ATTN_H2_VARIANT=$v rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_attn -o attn -- python $GRAFT_REPO_ROOT/tools/bench_attn.py $ARGS > /dev/null 2>&1
  DB=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_attn -name "*.db" | head -1)
  echo "variant $v: $(python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB | grep -E "h2_kernel|pack|rows" | awk -F, '{printf "%s %s | ", substr($1,22,14), $4}')"
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_attn
done
