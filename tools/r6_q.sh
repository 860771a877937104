cd $GRAFT_REPO_ROOT
TAG=r6_q timeout 1200 bash tools/prof_step.sh > /dev/null 2>&1
head -50 gpurun_out/r6_q_kernel_stats.csv | cut -c1-140
head -32 gpurun_out/r6_q_attrib.txt
