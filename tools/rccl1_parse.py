"""Kernel trace of tools/rccl1_step.py -> which stream / queue the RCCL kernels ran on, next to the marker fills launched on
the communication stream, and what ran concurrently with them (the ViT backward's kernels if the overlap works)."""
import csv
import glob
import sys
from collections import Counter, defaultdict

d = sys.argv[1]
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
if not rows:
    print("no kernel trace found under", d)
    sys.exit(1)
cols = rows[0].keys()
sid = "Stream_Id" if "Stream_Id" in cols else None
qid = "Queue_Id" if "Queue_Id" in cols else None
print("columns:", ", ".join(cols))
t0 = min(int(r["Start_Timestamp"]) for r in rows)


def key(r):
    return (r.get(qid, "?") if qid else "?", r.get(sid, "?") if sid else "?")


by = defaultdict(Counter)
for r in rows:
    by[key(r)][r["Kernel_Name"][:60]] += 1
print("\n(queue, stream) -> kernels (top 4 names, launches)")
for k, c in sorted(by.items(), key=lambda kv: -sum(kv[1].values())):
    print(" ", k, sum(c.values()), "launches:", "; ".join(f"{n} x{v}" for n, v in c.most_common(4)))
rccl = [r for r in rows if "nccl" in r["Kernel_Name"].lower() or "rccl" in r["Kernel_Name"].lower()]
fills = [r for r in rows if "fill" in r["Kernel_Name"].lower() and int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0) > 0]
print("\nRCCL kernels:", len(rccl), " distinct names:", sorted({r["Kernel_Name"][:80] for r in rccl}))
print("RCCL kernels on (queue, stream):", Counter(key(r) for r in rccl))
mk = Counter(key(r) for r in fills)
print("fill kernels on (queue, stream):", mk)
if rccl:
    print("\nlast step's collectives: start_us dur_us | kernels of OTHER streams running in that interval")
    last = rccl[-8:]
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        conc = Counter(o["Kernel_Name"][:50] for o in rows if key(o) != key(r) and int(o["Start_Timestamp"]) < e and int(o["End_Timestamp"]) > s)
        print("  %10.1f %8.1f | %s" % ((s - t0) / 1e3, (e - s) / 1e3, "; ".join(f"{n} x{v}" for n, v in conc.most_common(3)) or "-"))
