# SQ-side counters of one launch family: usage  pmc_sq.sh <script.py> <kernel substring>   (env passes through to the script)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcsq
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $R/gpurun_out/pmcsq --output-format csv -- python $R/$1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d $R/gpurun_out/pmcsq2 --output-format csv -- python $R/$1 > /dev/null 2>&1
python - "$2" <<'PY'
import csv, glob, os, sys, collections
R=os.environ["GRAFT_REPO_ROOT"]; pat=sys.argv[1]
for d in ("pmcsq", "pmcsq2"):
    acc=collections.defaultdict(dict); names={}
    for f in glob.glob(R+f"/gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"]:
                acc[int(row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"]); names[int(row["Dispatch_Id"])] = row["Kernel_Name"][:90]
    dur={}
    for f in glob.glob(R+f"/gpurun_out/{d}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            dur[int(row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    for k in sorted(acc)[-2:]:
        print(k, names[k], "us=%.1f" % dur.get(k, -1), {a: round(b) for a, b in sorted(acc[k].items())})
PY
