"""profiles/dominant_dispatches.json from the per-dispatch rows of the dominant launch (tools/rocpd_dispatches.py): what bench.py
quotes as roofline.rocprof_in_step, keyed to the kernel source it was measured on.
Usage: python tools/dispatch_record.py <dispatches.csv> <tag>"""
import csv
import hashlib
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = [r for r in csv.reader(open(sys.argv[1])) if r and r[0][0].isdigit()]
dur = [float(r[1]) for r in rows]
sha = hashlib.sha256(open(os.path.join(R, "semivl_amd", "csrc", "gemm_planes_impl.h"), "rb").read()).hexdigest()[:16]
print(json.dumps(dict(src_sha16=sha, M=32800, N=3072, K=768, workgroups=int(rows[0][2]), n=len(dur),
                      mean_us=round(sum(dur) / len(dur), 2), min_us=round(min(dur), 2), max_us=round(max(dur), 2),
                      source=f"profiles/{sys.argv[2]}_dominant_dispatches.csv (rocprofv3 --kernel-trace of `python bench.py --steps 2 "
                             f"--warmup 1`, every dispatch of gemm_x6p_kernel<2, 256, EPI_GELU> with 1548 workgroups)"), indent=1))
