# Same question as run13.sh with the tile width pinned to 256 (the cost model picks 128-wide tiles for small M): 60 / 120 / 252
# tiles of 256 x 256, one per CU, modes 0 (fp32 C) / 3 (planes) / 4 (pre-activation + planes).
cd $GRAFT_REPO_ROOT
B=tools/micro/x6p_bench
export SVL_PLANES_TILE=256
for mode in 0 3 4; do for M in 1280 2560 5376; do for K in 768 3072; do
  echo -n "mode $mode M $M K $K: "; timeout 120 $B $M 3072 $K 40 $mode | tail -1
done; done; done
