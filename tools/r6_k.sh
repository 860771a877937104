cd $GRAFT_REPO_ROOT
TAG=r6_k_ade timeout 1200 bash tools/prof_step.sh --config ade > /dev/null 2>&1
head -45 gpurun_out/r6_k_ade_kernel_stats.csv | cut -c1-140
head -30 gpurun_out/r6_k_ade_attrib.txt
