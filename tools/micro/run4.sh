cd $GRAFT_REPO_ROOT
B=tools/micro/x6p_bench
export SVL_X6P_NOSPLIT=1
for M in 5376 2816 1280 256; do
 for mode in 0 1 3; do
  for dbg in 0 16; do
   echo -n "M $M mode $mode dbg $dbg: "; SVL_X6P_DBG=$dbg timeout 120 $B $M 3072 768 50 $mode | tail -1
  done
 done
done
