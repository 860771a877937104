# Is the epilogue of the planes GEMM bound per CU (store path) or per chip (HBM writes)?  One round of tiles on 60 / 252
# CUs and the full shape, three K (slope = k-loop, intercept = prologue + epilogue), three output modes.
cd $GRAFT_REPO_ROOT
B=tools/micro/x6p_bench
for mode in 0 3 4; do for M in 1280 5376 32768; do for K in 768 1536 3072; do
  echo -n "mode $mode M $M K $K: "; timeout 120 $B $M 3072 $K 30 $mode | tail -1
done; done; done
