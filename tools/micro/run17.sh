cd $GRAFT_REPO_ROOT
B=tools/micro/x6p_phases
for mode in 0 4; do for np in 2 3; do timeout 120 $B 32768 3072 768 $mode $np; done; done
timeout 120 $B 32768 768 3072 0 2
timeout 120 $B 5376 3072 768 4 2
NO_FAST=1 timeout 120 $B 32768 3072 768 4 2
