# PMC counters of one svl_gemm_planes_f32 launch (the C-ABI harness tools/micro/x6p_bench): usage  pmc_x6p.sh M N K mode
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="${1:-32768} ${2:-3072} ${3:-768} 3 ${4:-0}"
export X6P_ARGS="$ARGS"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcx_$tag --output-format csv -- $R/tools/micro/x6p_bench $ARGS > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmcx_*/**/*counter_collection.csv", recursive=True)):
    acc=collections.OrderedDict()
    for row in csv.DictReader(open(f)):
        if "gemm_x6p" not in row["Kernel_Name"]: continue
        acc[(int(row["Dispatch_Id"]),row["Counter_Name"])]=float(row["Counter_Value"])
    if not acc: continue
    last=max(k[0] for k in acc)
    for (d,c),v in acc.items():
        if d==last: print(c, v)
# record for bench.py (roofline.traffic): keyed to the kernel source it was measured on
import hashlib, json, sys
vals={}
for f in sorted(glob.glob(R+"/gpurun_out/pmcx_*/**/*counter_collection.csv", recursive=True)):
    acc={}
    for row in csv.DictReader(open(f)):
        if "gemm_x6p" in row["Kernel_Name"]: acc[(int(row["Dispatch_Id"]),row["Counter_Name"])]=float(row["Counter_Value"])
    if acc:
        last=max(k[0] for k in acc)
        for (d,c),v in acc.items():
            if d==last: vals[c]=v
a=os.environ.get("X6P_ARGS","32768 3072 768 3 0").split()
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    sha=hashlib.sha256(open(R+"/semivl_amd/csrc/gemm_planes_impl.h","rb").read()).hexdigest()[:16]
    fb, wb = vals["FETCH_SIZE"]*1024*2, vals["WRITE_SIZE"]*1024
    clk = vals.get("GRBM_GUI_ACTIVE")
    rec=dict(src_sha16=sha, fmt=os.environ.get("X6P_FMT", "0"), M=int(a[0]), N=int(a[1]), K=int(a[2]), mode=int(a[4]), fetch_bytes=fb, write_bytes=wb, traffic_bytes=fb+wb,
             counters=vals,
             note="FETCH_SIZE (KB, doubled: gfx950 under-reads 16 B/lane loads by 2x, MI355X_MICROARCH.md) + WRITE_SIZE of the last "
                  "gemm_x6p_kernel launch of tools/micro/x6p_bench (mode %s: 1 = bias + GELU + pre-activation + planes out + fp32 C, 4 = the same without C -- FFN-1 as the training step launches it)" % a[4] + ", separate "
                  "rocprofv3 --pmc passes (tools/pmc_x6p.sh).  FETCH_SIZE counts L2 -> fabric requests, Infinity-Cache hits included: "
                  "the B panels (14 MB) are re-read by every XCD once per round of its tiles and are served from the Infinity Cache")
    json.dump(rec, open(R+"/gpurun_out/pmc_x6p_traffic.json","w"), indent=1)
    print("wrote gpurun_out/pmc_x6p_traffic.json", rec["traffic_bytes"]/1e9, "GB")
for f in sorted(glob.glob(R+"/gpurun_out/pmcx_SQ_WAVE_CYCLES/**/*kernel_trace.csv", recursive=True)):
    rows=[r for r in csv.DictReader(open(f)) if "gemm_x6p" in r["Kernel_Name"]]
    for r in rows[-2:]: print("duration_us", (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, "VGPR", r.get("VGPR_Count"), "LDS", r.get("LDS_Block_Size"))
PY
rm -rf $R/gpurun_out/pmcx_*
