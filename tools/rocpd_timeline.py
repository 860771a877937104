"""GPU occupancy of a rocprofv3 --kernel-trace run (rocpd sqlite): union-busy time, idle gaps, per-queue busy time.
Usage: python tools/rocpd_timeline.py <db> [first_fraction last_fraction]   (fractions of the dispatch list to analyse)"""
import sqlite3
import sys

db = sys.argv[1]
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in cols else None
scol = "stream_id" if "stream_id" in cols else None
sel = "d.start, d.end, s.kernel_name" + (f", d.{qcol}" if qcol else ", 0") + (f", d.{scol}" if scol else ", 0")
rows = c.execute(f"select {sel} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                 "order by d.start").fetchall()
n = len(rows)
rows = rows[int(f0 * n):int(f1 * n)]
t0, t1 = rows[0][0], max(r[1] for r in rows)
span = t1 - t0
busy = 0
cur_s, cur_e = rows[0][0], rows[0][1]
gaps = []
for s, e, name, q, st in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, name))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"dispatches {len(rows)}  span {span / 1e6:.2f} ms  union-busy {busy / 1e6:.2f} ms  idle {100.0 * (span - busy) / span:.1f} %  "
      f"sum of kernel time {sum(r[1] - r[0] for r in rows) / 1e6:.2f} ms")
per = {}
for s, e, name, q, st in rows:
    k = (q, st)
    per[k] = per.get(k, 0) + e - s
for k, v in sorted(per.items(), key=lambda kv: -kv[1]):
    print(f"  queue {k[0]} stream {k[1]}: busy {v / 1e6:.2f} ms")
edges = [1e3, 5e3, 2e4, 1e5, 1e6, 1e12]
hist = [0] * len(edges)
tot = [0] * len(edges)
for g, _ in gaps:
    for i, ed in enumerate(edges):
        if g <= ed:
            hist[i] += 1
            tot[i] += g
            break
print("idle gaps (count, total ms) by length: " + ", ".join(
    f"<={int(ed / 1e3)}us: {h} / {t / 1e6:.2f}" for ed, h, t in zip(edges[:-1], hist, tot)) + f", longer: {hist[-1]} / {tot[-1] / 1e6:.2f}")
big = sorted(gaps, reverse=True)[:12]
print("longest gaps (us, next kernel):")
for g, name in big:
    print(f"  {g / 1e3:.1f}  {name[:90]}")
