# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of one launch: usage  pmc_traffic.sh <script.py> <kernel substring>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmct_$c --output-format csv -- python $R/$1 > /dev/null 2>&1
done
python - "$2" <<'PY'
import csv, glob, os, sys
R=os.environ["GRAFT_REPO_ROOT"]; pat=sys.argv[1]
res={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    rows=[]
    for f in glob.glob(R+f"/gpurun_out/pmct_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"] and row["Counter_Name"]==c: rows.append((int(row["Dispatch_Id"]), float(row["Counter_Value"]), row["Kernel_Name"][:70]))
    rows.sort()
    if rows: print(c, "KB (last launch):", rows[-1][1], rows[-1][2])
    res[c] = rows[-1][1] if rows else None
# record for bench.py (roofline.dominant_launch.traffic): keyed to the kernel source it was measured on
emu = os.environ.get("SVL_GEMM_EMU", "0").strip() or "0"
if ((pat == "gemm_kernel" and emu == "0") or (pat == "gemm_bf16x" and emu == "6")) and all(res.get(c) for c in ("FETCH_SIZE", "WRITE_SIZE")):
    import hashlib, json
    sha = hashlib.sha256(open(R + "/semivl_amd/csrc/gemm.hip", "rb").read()).hexdigest()[:16]
    fetch_b, write_b = res["FETCH_SIZE"] * 1024 * 2, res["WRITE_SIZE"] * 1024     # KB units; FETCH_SIZE x2 on gfx950 (16 B/lane reads)
    rec = dict(gemm_hip_sha16=sha, M=32800, N=int(os.environ.get("ONE_N", 3072)), K=int(os.environ.get("ONE_K", 768)),
               fetch_bytes=fetch_b, write_bytes=write_b, traffic_bytes=fetch_b + write_b,
               note="FETCH_SIZE (KB, doubled: gfx950 under-reads 16 B/lane loads by 2x, MI355X_MICROARCH.md) + WRITE_SIZE of the "
                    "last %s launch of tools/one_gemm.py, separate rocprofv3 --pmc passes (tools/pmc_traffic.sh)" % pat)
    os.makedirs(R + "/gpurun_out", exist_ok=True)
    name = "pmc_gemm_traffic.json" if emu == "0" else "pmc_gemm_traffic_bf16x6.json"
    json.dump(rec, open(R + "/gpurun_out/" + name, "w"), indent=1)
    print("wrote gpurun_out/" + name + ":", rec["traffic_bytes"] / 1e9, "GB")
PY
rm -rf $R/gpurun_out/pmct_*
