cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU -d $R/gpurun_out/pmca --output-format csv -- python $R/tools/one_attn.py > $R/gpurun_out/pmca.log 2>&1; tail -5 $R/gpurun_out/pmca.log
python - <<'PY'
import csv, glob, os, collections, re
R=os.environ["GRAFT_REPO_ROOT"]
acc=collections.OrderedDict()
for f in glob.glob(R+"/gpurun_out/pmca/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m=re.search(r"(attn_\w+)", row["Kernel_Name"]); k=m.group(1) if m else None
        if not k: continue
        acc[(k,row["Counter_Name"])]=float(row["Counter_Value"])   # last launch wins
ks=sorted({k for k,_ in acc})
for k in ks:
    g=acc.get((k,"GRBM_GUI_ACTIVE"),0)/8; m=acc.get((k,"SQ_VALU_MFMA_BUSY_CYCLES"),0)/1024
    if g: print(f"{k:28s} cycles {g:10.0f}  MFMA-busy {100*m/g:5.1f} %  VALU/MFMA {acc.get((k,'SQ_INSTS_VALU'),0)/max(acc.get((k,'SQ_INSTS_MFMA'),1),1):5.2f}")
PY
rm -rf $R/gpurun_out/pmca
