# PMC counters of one svl_gemm_planes_f32 launch (the C-ABI harness tools/micro/x6p_bench): usage  pmc_x6p.sh M N K mode
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="${1:-32768} ${2:-3072} ${3:-768} 3 ${4:-0}"
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
for f in sorted(glob.glob(R+"/gpurun_out/pmcx_SQ_WAVE_CYCLES/**/*kernel_trace.csv", recursive=True)):
    rows=[r for r in csv.DictReader(open(f)) if "gemm_x6p" in r["Kernel_Name"]]
    for r in rows[-2:]: print("duration_us", (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, "VGPR", r.get("VGPR_Count"), "LDS", r.get("LDS_Block_Size"))
PY
rm -rf $R/gpurun_out/pmcx_*
