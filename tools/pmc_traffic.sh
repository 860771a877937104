# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of one launch: usage  pmc_traffic.sh <script.py> <kernel substring>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmct_$c --output-format csv -- python $R/$1 > /dev/null 2>&1
done
python - "$2" <<'PY'
import csv, glob, os, sys
R=os.environ["GRAFT_REPO_ROOT"]; pat=sys.argv[1]
for c in ("FETCH_SIZE","WRITE_SIZE"):
    rows=[]
    for f in glob.glob(R+f"/gpurun_out/pmct_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"] and row["Counter_Name"]==c: rows.append((int(row["Dispatch_Id"]), float(row["Counter_Value"]), row["Kernel_Name"][:70]))
    rows.sort()
    if rows: print(c, "KB (last launch):", rows[-1][1], rows[-1][2])
PY
rm -rf $R/gpurun_out/pmct_*
