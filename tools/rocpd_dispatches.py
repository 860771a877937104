"""Per-dispatch rows of ONE kernel from a rocprofv3 --kernel-trace run (rocpd sqlite): the dominant launch's duration as
rocprof sees it, one row per launch (not the family average of the stats table).
Usage: python tools/rocpd_dispatches.py <db> <kernel name substring> [grid workgroups]"""
import sqlite3
import sys

db, pat = sys.argv[1], sys.argv[2]
grid = int(sys.argv[3]) if len(sys.argv) > 3 else None
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
gcol = next((k for k in ("grid_size_x", "grid_x", "grid_size") if k in cols), None)
wcol = next((k for k in ("workgroup_size_x", "workgroup_x", "workgroup_size") if k in cols), None)
sel = "d.start, d.end, s.kernel_name" + (f", d.{gcol}" if gcol else ", 0") + (f", d.{wcol}" if wcol else ", 1")
rows = c.execute(f"select {sel} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                 "where s.kernel_name like ? order by d.start", (f"%{pat}%",)).fetchall()
out = []
for s_, e_, name, g, w in rows:
    wgs = (g // w) if (g and w) else None
    if grid is not None and wgs is not None and wgs != grid:
        continue
    out.append((s_, (e_ - s_) / 1e3, wgs))
print(f"# {len(out)} dispatches of *{pat}*" + (f" with {grid} workgroups" if grid else "") +
      (f": mean {sum(o[1] for o in out) / len(out):.2f} us, min {min(o[1] for o in out):.2f}, max {max(o[1] for o in out):.2f}" if out else ""))
print("start_ns,duration_us,workgroups")
for s_, d_, wgs in out:
    print(f"{s_},{d_:.2f},{wgs}")
