cd $GRAFT_REPO_ROOT
python tools/dbg_batchinv.py 2>&1 | tail -8
