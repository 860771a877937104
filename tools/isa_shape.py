"""Run-length 'shape' of a kernel's instruction stream (M = MFMA, v = VALU, r/w = LDS read/write, G/S = global load/
store, W = s_waitcnt, B = barrier, s = scalar).  Usage: python tools/isa_shape.py file.s kernel_substring"""
import sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
a = s.index(key + ":") if (key + ":") in s else s.index(":", s.index(key))
a = s.rfind("\n", 0, a) + 1
b = s.index(".Lfunc_end", a)
seq = []
for ln in s[a:b].split("\n"):
    t = ln.strip()
    if t.startswith(".LBB"):
        seq.append(("LBL", t)); continue
    if not t or t[0] in ";." or t.endswith(":"):
        continue
    op = t.split()[0]
    c = ("M" if op.startswith("v_mfma") else "r" if op.startswith("ds_read") or op.startswith("ds_load") else
         "w" if op.startswith("ds_write") or op.startswith("ds_store") else
         "G" if op.startswith(("global_load", "buffer_load")) else "S" if op.startswith("global_store") else
         "A" if op.startswith("v_accvgpr") else "v" if op.startswith("v_") else "W" if op.startswith("s_waitcnt") else
         "B" if op.startswith("s_barrier") else "J" if op.startswith(("s_cbranch", "s_branch")) else
         "n" if op.startswith("s_nop") else "s")
    seq.append((c, op))
out, prev, cnt = [], None, 0
for c, op in seq:
    if c == "LBL":
        if prev: out.append(f"{prev}{cnt}")
        out.append("\n" + op + " "); prev, cnt = None, 0
        continue
    if c == prev: cnt += 1
    else:
        if prev: out.append(f"{prev}{cnt}")
        prev, cnt = c, 1
if prev: out.append(f"{prev}{cnt}")
print(" ".join(out))
