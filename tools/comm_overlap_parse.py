"""From a rocprofv3 --kernel-trace (rocpd sqlite) of tools/comm_overlap_step.py: the injected collective's kernels
(permute_rows_kernel: only that launch uses it in a VOC step) must sit on a hardware queue of their own and run while other kernels of the
step are in flight.  Prints a record and exits 1 when either does not hold.  Usage: python tools/comm_overlap_parse.py <db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in cols else None
scol = "stream_id" if "stream_id" in cols else None
sel = "d.start, d.end, s.kernel_name" + (f", d.{qcol}" if qcol else ", 0") + (f", d.{scol}" if scol else ", 0")
rows = c.execute(f"select {sel} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                 "order by d.start").fetchall()
comm = [r for r in rows if "permute_rows_kernel" in r[2]]
# (clock_probe_kernel = the reducer's own construction-time queue probe on its candidate streams: not a kernel of the step)
rest = [r for r in rows if "permute_rows_kernel" not in r[2] and "clock_probe_kernel" not in r[2]]
assert comm, "no collective kernels in the trace"
comm_q = {r[3] for r in comm}
comm_s = {r[4] for r in comm}
# the x2 kernels that follow on the same stream belong to the collective too
comm_stream_rows = [r for r in rest if r[4] in comm_s] if scol else []
others = [r for r in rest if not (scol and r[4] in comm_s)]
other_q = {r[3] for r in others}
shared = comm_q & other_q
# overlap: for each collective kernel, the part of its interval during which at least one other kernel is running
ev = sorted([(r[0], 1) for r in others] + [(r[1], -1) for r in others])
covered, depth, last, spans = 0, 0, None, []
for t, d in ev:
    if depth > 0 and last is not None and t > last:
        spans.append((last, t))
    depth += d
    last = t
merged = []
for s_, e_ in spans:
    if merged and s_ <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e_)
    else:
        merged.append([s_, e_])
tot, ov = 0, 0
for s_, e_, *_ in comm:
    tot += e_ - s_
    for ms, me in merged:
        lo, hi = max(s_, ms), min(e_, me)
        if hi > lo:
            ov += hi - lo
frac = ov / max(tot, 1)
print(f"collective kernels (permute_rows_kernel, bucket-sized): {len(comm)} dispatches, {tot / 1e6:.3f} ms in total, on queue(s) {sorted(comm_q)} "
      f"stream(s) {sorted(comm_s)}; + {len(comm_stream_rows)} follow-up kernels on the same stream")
print(f"the step's other kernels: {len(others)} dispatches on queue(s) {sorted(other_q)}")
print(f"queues shared between the collective and the step's kernels: {sorted(shared) if shared else 'none'}")
if shared:
    names = {}
    for r in others:
        if r[3] in shared:
            k = (r[4], r[2][:60])
            names[k] = names.get(k, 0) + 1
    for (st_, nm_), n_ in sorted(names.items(), key=lambda kv: -kv[1])[:8]:
        print(f"    on a shared queue: stream {st_}  x{n_}  {nm_}")
print(f"fraction of the collective kernels' time during which other kernels of the step were running: {frac:.3f}")
ok = (not shared) and frac > 0.8
print("RESULT:", "ok -- the collective owns its hardware queue and runs under the step's kernels" if ok else "FAILED")
sys.exit(0 if ok else 1)
