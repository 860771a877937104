# SQ counters of the tiled 3x3 weight-gradient kernel (bf16x6 when EMU=6): usage  EMU=6 ONE_C=128 ONE_CO=64 bash tools/pmc_wgrad.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ONE_MODE=wgrad SVL_GEMM_EMU=${EMU:-0}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcw_$tag --output-format csv -- python $R/tools/one_conv.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmcw_*/**/*counter_collection.csv", recursive=True)):
    acc=collections.OrderedDict()
    for row in csv.DictReader(open(f)):
        if "wgrad_tiled" not in row["Kernel_Name"]: continue
        acc[(row["Dispatch_Id"],row["Counter_Name"])]=(float(row["Counter_Value"]), row["Kernel_Name"][:70])
    if not acc: continue
    last=max(int(k[0]) for k in acc)
    for (d,c),v in acc.items():
        if int(d)==last: print(c, v[0], v[1] if c=="SQ_WAVE_CYCLES" else "")
PY
rm -rf $R/gpurun_out/pmcw_*
