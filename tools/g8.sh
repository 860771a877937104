cd $GRAFT_REPO_ROOT
TAG=r5_b bash tools/prof_step.sh > /dev/null 2>&1
python tools/shape_table.py voc > gpurun_out/r5_b_shapes_voc.txt 2>/dev/null
