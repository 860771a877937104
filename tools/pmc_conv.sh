cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcc_$tag --output-format csv -- python $R/tools/one_conv.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmcc_*/**/*counter_collection.csv", recursive=True)):
    acc=collections.OrderedDict()
    for row in csv.DictReader(open(f)):
        if "gemm_kernel" not in row["Kernel_Name"]: continue
        acc[(row["Dispatch_Id"],row["Counter_Name"])]=(float(row["Counter_Value"]), row["Kernel_Name"][:60])
    last=max(int(k[0]) for k in acc)
    for (d,c),v in acc.items():
        if int(d)==last: print(c, v[0], v[1] if c.endswith("CYCLES") and "WAVE" in c else "")
PY
