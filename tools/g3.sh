cd $GRAFT_REPO_ROOT
for n in 60 256; do for nt in 0 1; do tools/micro/storebw $n 640 $nt; done; done
tools/micro/storebw 256 256 0; tools/micro/storebw 60 256 0
timeout 900 python -m pytest tests/test_multiproc_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "rccl or swapped" -s 2>&1 | tail -15
TAG=r5_a bash tools/rccl1_timeline.sh
