cd $GRAFT_REPO_ROOT
B=tools/micro/x6p_bench
for dbg in 0 1 2 3 4 5 7 8 16 17 24; do
  echo "== dbg $dbg"
  SVL_X6P_DBG=$dbg timeout 120 $B 8192 8192 8192 5 0 | tail -1
  SVL_X6P_DBG=$dbg timeout 120 $B 32768 3072 768 20 0 | tail -1
  SVL_X6P_DBG=$dbg timeout 120 $B 32768 768 3072 20 0 | tail -1
done
