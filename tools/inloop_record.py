"""profiles/inloop_dispatches.json: mean in-step kernel durations of the fused-attention launches (the `larger_shapes` of
bench.py's roofline object) from a rocprofv3 --kernel-trace run of `python bench.py` (rocpd sqlite), keyed to the kernel
source they were measured on.  Only the launches over 2 x batch images (the grad-carrying passes: 12 x 2 B x ceil(T / 256)
workgroups) are counted.   Usage: python tools/inloop_record.py <db> <tag> [batch]"""
import hashlib
import json
import os
import sqlite3
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
db, tag = sys.argv[1], sys.argv[2]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
gcol = next((k for k in ("grid_size_x", "grid_x", "grid_size") if k in cols), None)
wcol = next((k for k in ("workgroup_size_x", "workgroup_x", "workgroup_size") if k in cols), None)


def mean_of(pat, wgs=None):
    rows = c.execute(f"select d.start, d.end, d.{gcol}, d.{wcol} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                     "on d.kernel_id = s.id where s.kernel_name like ?", (f"%{pat}%",)).fetchall()
    d = [(e - s) / 1e3 for s, e, g, w in rows if wgs is None or (g // w) == wgs]
    return dict(mean_us=round(sum(d) / len(d), 2), n=len(d)) if d else None


z = 2 * batch * 12
out = dict(attn_h2_sha16=hashlib.sha256(open(os.path.join(R, "semivl_amd", "csrc", "attn_h2.hip"), "rb").read()).hexdigest()[:16],
           source=f"profiles/{tag}_kernel_stats_bs16_bf16x6.csv's rocprofv3 run (python bench.py --steps 2 --warmup 1), launches over "
                  f"{2 * batch} images",
           kernels={
               "fwd_h2": dict(grid=mean_of("attn_fwd_h2_kernel", 4 * z), pack=mean_of("attn_pack_kernel"),
                              tail=mean_of("attn_fwd_tail_h2_kernel", z)),
               "bwd_h2": dict(dkv=mean_of("attn_dkv_h2_kernel", 4 * z), dq=mean_of("attn_dq_h2_kernel", 4 * z),
                              ld=mean_of("attn_ld_kernel"), dkv_tail=mean_of("attn_dkv_tail_h2_kernel", z),
                              dq_tail=mean_of("attn_dq_tail_h2_kernel", z))})
print(json.dumps(out, indent=1))
