cd $GRAFT_REPO_ROOT
B=tools/micro/x6p_bench
for args in "300 208 64 2 1" "513 400 128 2 2" "20808 512 768 5 0" "20816 2304 768 5 0" "8200 768 768 5 2" "32800 3072 768 20 0" "32800 3072 768 20 1" "32800 768 768 20 2" "32800 768 3072 20 2" "32800 2304 768 20 0" "16400 3072 768 20 1" "8192 8192 8192 5 0"; do echo -n "$args: "; timeout 120 $B $args | tr '\n' ' ' | sed 's/max |err|//; s/(max.*bad/bad/'; echo; done
echo "== no persist"
for args in "32800 3072 768 20 0" "32800 3072 768 20 1" "32800 768 768 20 2" "32800 768 3072 20 2" "32800 2304 768 20 0"; do echo -n "$args: "; SVL_X6P_NO_PERSIST=1 timeout 120 $B $args | tail -1; done
