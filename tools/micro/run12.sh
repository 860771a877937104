# FETCH_SIZE of one FFN-1 launch (harness mode 4) per panel width of the tile order
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pw in 0 2 3 4 6; do
  rm -rf $R/gpurun_out/pmcp_$pw
  SVL_PLANES_PANEL=$pw timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmcp_$pw --output-format csv -- $R/tools/micro/x6p_bench ${X6P_SHAPE:-32800 3072 768} 3 ${X6P_MODE:-4} > /dev/null 2>&1
  python3 - $pw <<'PY'
import csv, glob, os, sys
R=os.environ["GRAFT_REPO_ROOT"]; pw=sys.argv[1]
for f in glob.glob(R+f"/gpurun_out/pmcp_{pw}/**/*counter_collection.csv", recursive=True):
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_x6p" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
    if v: print(f"panel {pw}: FETCH_SIZE x 2 = {v[-1]*2048/1e9:.3f} GB")
PY
  rm -rf $R/gpurun_out/pmcp_$pw
done
