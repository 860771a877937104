# which stream / hardware queue do the RCCL kernels of the gradient all-reduce land on?  (one rank, one GPU)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/rccl1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rccl1 -- python $R/tools/rccl1_step.py > $R/gpurun_out/rccl1_run.log 2>&1
{ grep RCCL1_ $R/gpurun_out/rccl1_run.log; python $R/tools/rccl1_parse.py $R/gpurun_out/rccl1; } > $R/gpurun_out/${TAG:-r5}_rccl1_timeline.txt 2>&1
rm -rf $R/gpurun_out/rccl1
tail -40 $R/gpurun_out/${TAG:-r5}_rccl1_timeline.txt
