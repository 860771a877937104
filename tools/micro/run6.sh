cd $GRAFT_REPO_ROOT
B=tools/micro/x6p_bench
for t in 128 192 256; do
  echo "== tile $t"
  for args in "32800 768 768 20 2" "32800 768 3072 20 2" "32800 2304 768 20 0" "32800 768 2304 20 0" "32800 3072 768 20 1" "16400 768 3072 20 2" "16400 2304 768 20 0" "300 208 64 2 1" "2600 2304 768 5 1"; do echo -n "$args: "; SVL_PLANES_TILE=$t timeout 120 $B $args | tr '\n' ' ' | sed 's/max |err|//; s/(max.*bad/bad/'; echo; done
done
