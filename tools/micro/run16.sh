# fast epilogue (no load behind a store) against the generic one, both operand formats, the ViT's shapes + error check
cd $GRAFT_REPO_ROOT
B=tools/micro/x6p_bench
for args in "300 208 64 2 1" "513 400 128 2 2" "20808 512 768 5 0" "8200 768 768 10 2" "32800 3072 768 20 0" "32800 3072 768 20 4" "32800 3072 768 20 3" "32800 768 768 20 2" "32800 768 3072 20 2" "32800 2304 768 20 0" "32800 768 2304 20 0" "5376 3072 768 40 4" "1280 3072 768 40 4"; do
  for f in 1 0; do for nf in 0 1; do echo "fmt $f nofast $nf: $(SVL_PLANES_NO_FAST_EPI=$nf X6P_FMT=$f timeout 120 $B $args | tr '\n' ' ')"; done; done
done
