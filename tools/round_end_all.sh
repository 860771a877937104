# harness on the shapes that matter + the planes tests + the evidence part 2 (one box)
cd $GRAFT_REPO_ROOT
B=tools/micro/x6p_bench
for args in "300 208 64 2 1" "20808 512 768 5 0" "20816 2304 768 5 0" "32800 3072 768 20 0" "32800 3072 768 20 1" "32800 768 3072 20 2"; do echo -n "$args: "; timeout 120 $B $args | tr '\n' ' ' | sed 's/max |err|//; s/(max.*bad/bad/'; echo; done > gpurun_out/x6p_11.log 2>&1
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "planes" > gpurun_out/pytest_planes.log 2>&1
tail -3 gpurun_out/pytest_planes.log
bash tools/round_end_part2.sh
