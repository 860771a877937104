# rocprofv3 evidence of the reducer's stream-ordered branch on one GPU (usage: TAG=r6_x bash tools/comm_overlap_trace.sh)
T=${TAG:-r6}
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_comm_$T
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_comm_$T -o comm -- python $R/tools/comm_overlap_step.py > $R/gpurun_out/${T}_comm_overlap.txt 2>/dev/null
cd $R; DB=$(find gpurun_out/prof_comm_$T -name "*.db" | head -1)
python tools/comm_overlap_parse.py $DB >> gpurun_out/${T}_comm_overlap.txt; rc=$?
rm -rf gpurun_out/prof_comm_$T
cat gpurun_out/${T}_comm_overlap.txt
exit $rc
