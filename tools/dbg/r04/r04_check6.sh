cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python tools/dbg/cumask_side.py 2>&1 | tail -9
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
