# HBM-side traffic (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes) of one ce_fused_kernel launch at the bench shape
# -> gpurun_out/pmc_ce_traffic.json (copy to profiles/; bench.py quotes it as roofline_hbm.traffic)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmcce_$c --output-format csv -- python $R/tools/one_ce.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os, json
R=os.environ["GRAFT_REPO_ROOT"]
res={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    rows=[]
    for f in glob.glob(R+f"/gpurun_out/pmcce_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "ce_fused" in row["Kernel_Name"] and row["Counter_Name"]==c: rows.append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    rows.sort()
    res[c]=rows[-1][1] if rows else None
    print(c, "KB (last launch):", res[c])
if all(res.values()):
    fb, wb = res["FETCH_SIZE"]*1024*2, res["WRITE_SIZE"]*1024
    B,N,S=int(os.environ.get("ONE_B",16)),int(os.environ.get("ONE_NCLS",21)),int(os.environ.get("ONE_S",512))
    rec=dict(B=B,N=N,HW=S*S,fetch_bytes=fb,write_bytes=wb,traffic_bytes=fb+wb,algorithmic_bytes=float(B*S*S*(12*N+40)),
             note="FETCH_SIZE (KB, doubled: gfx950 under-reads 16 B/lane loads by 2x, MI355X_MICROARCH.md) + WRITE_SIZE of the last "
                  "ce_fused_kernel launch of tools/one_ce.py, separate rocprofv3 --pmc passes (tools/pmc_ce.sh)")
    json.dump(rec, open(R+"/gpurun_out/pmc_ce_traffic.json","w"), indent=1)
    print("wrote gpurun_out/pmc_ce_traffic.json:", rec["traffic_bytes"]/1e9, "GB vs algorithmic", rec["algorithmic_bytes"]/1e9)
PY
rm -rf $R/gpurun_out/pmcce_*
