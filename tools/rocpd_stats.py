"""Per-kernel summary (calls, total/avg duration, %) from a rocprofv3 rocpd sqlite database (--kernel-trace)."""
import re
import sqlite3
import sys

db = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 9   # every kernel unless a cut is asked for
c = sqlite3.connect(db)
rows = c.execute("""select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
                    from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                    group by s.kernel_name order by 3 desc""").fetchall()
tot = sum(r[2] for r in rows)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    return n[:110]


print(f"# rocprofv3 --kernel-trace summary: {len(rows)} kernels, total GPU kernel time {tot / 1e6:.2f} ms")
print("name,calls,total_ms,avg_us,min_us,max_us,percent")
for n, k, t, mn, mx in rows[:top]:
    print(f"\"{short(n)}\",{k},{t / 1e6:.3f},{t / k / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100.0 * t / tot:.2f}")
