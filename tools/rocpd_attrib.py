"""Wall-time attribution of a rocprofv3 --kernel-trace run (rocpd sqlite): for every instant of the analysed window, is a
matrix-pipe kernel running (GEMM / conv / attention), only bandwidth-type kernels, or nothing?  The time WITHOUT an MFMA
kernel in flight is what fusing or hiding the normalisation / elementwise / pack passes can recover at most.
Usage: python tools/rocpd_attrib.py <db> [first_fraction last_fraction]"""
import re
import sqlite3
import sys

db = sys.argv[1]
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
c = sqlite3.connect(db)
rows = c.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                 "on d.kernel_id = s.id order by d.start").fetchall()
n = len(rows)
rows = rows[int(f0 * n):int(f1 * n)]
MFMA = re.compile(r"gemm_x6p|gemm_bf16x|gemm_kernel|gemm_shortk|shortk_x6|attn_(fwd|bwd|dq_h2|dkv_h2)|conv3x3|seqattn")
ev = []
for s, e, name in rows:
    m = 1 if MFMA.search(name) and "rows_kernel" not in name else 0
    ev.append((s, 1, m, name))
    ev.append((e, -1, m, name))
ev.sort(key=lambda x: (x[0], x[1]))
t_prev = ev[0][0]
act_m = act_o = 0
tm = to = ti = 0
solo = {}
names_active = {}
for t, d, m, name in ev:
    dt = t - t_prev
    if dt > 0:
        if act_m > 0:
            tm += dt
        elif act_o > 0:
            to += dt
            for k in names_active:
                if names_active[k] > 0:
                    solo[k] = solo.get(k, 0) + dt / sum(1 for v in names_active.values() if v > 0)
        else:
            ti += dt
    t_prev = t
    if m:
        act_m += d
    else:
        act_o += d
        key = re.sub(r"\(anonymous namespace\)::|void ", "", name)[:60]
        names_active[key] = names_active.get(key, 0) + d
span = ev[-1][0] - ev[0][0]
print(f"window {span / 1e6:.1f} ms: MFMA kernel in flight {tm / 1e6:.1f} ms ({100 * tm / span:.1f} %), only non-MFMA kernels "
      f"{to / 1e6:.1f} ms ({100 * to / span:.1f} %), idle {ti / 1e6:.1f} ms ({100 * ti / span:.1f} %)")
print("non-MFMA kernels by wall time WITHOUT an MFMA kernel in flight (ms):")
for k, v in sorted(solo.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {v / 1e6:8.2f}  {k}")
