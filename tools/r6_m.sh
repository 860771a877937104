cd $GRAFT_REPO_ROOT
TAG=r6_m_ade timeout 1200 bash tools/prof_step.sh --config ade > /dev/null 2>&1
grep -E "conv3x3_dil|gemm_bf16x_kernel|shortk" gpurun_out/r6_m_ade_kernel_stats.csv | cut -c1-150
head -12 gpurun_out/r6_m_ade_kernel_stats.csv | cut -c1-140
