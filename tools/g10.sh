cd $GRAFT_REPO_ROOT
for f in h2 b3; do echo "== SVL_ATTN_FMT=$f"; SVL_ATTN_FMT=$f timeout 300 python tools/dbg_attn_err.py 2>&1 | grep -v Warn; done
