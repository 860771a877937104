cd $GRAFT_REPO_ROOT
B=tools/micro/x6p_bench
for args in "32800 3072 768 20 4" "32800 3072 768 20 0" "32800 768 3072 20 2" "32800 2304 768 20 0" "32800 768 768 20 2"; do
  for ds in 0 50 57 65; do echo "desync $ds: $(SVL_PLANES_DESYNC=$ds X6P_FMT=1 timeout 120 $B $args | tr '\n' ' ' | cut -c1-250)"; done
done
