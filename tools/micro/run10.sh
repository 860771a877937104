# sustained shader clock under the planes GEMM (probe waves on a second stream)
cd $GRAFT_REPO_ROOT
for args in "32800 3072 768 40 0" "32800 3072 768 40 1" "8192 8192 8192 6 0" "32800 768 3072 40 2"; do echo "== $args"; X6P_CLOCK=1 timeout 300 tools/micro/x6p_bench $args | tail -4; done
