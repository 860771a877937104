cd $GRAFT_REPO_ROOT
B=tools/micro/x6p_bench
export SVL_X6P_NOSPLIT=1
for M in 5376 32768; do
 for K in 256 768 1536 3072 6144; do
  for dbg in 16 17 23; do
   echo -n "M $M K $K dbg $dbg: "; SVL_X6P_DBG=$dbg timeout 120 $B $M 3072 $K 30 0 | tail -1
  done
 done
done
