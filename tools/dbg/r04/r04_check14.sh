set -x
cd $GRAFT_REPO_ROOT
echo "== A: as shipped (write-through + bare wait, no acquire)"; timeout 900 python tools/dbg/tail_stress.py 400
echo "== B: + acquire"; COVER_LIB_PATH=$PWD/build_dbg/libcover_tailB.so timeout 900 python tools/dbg/tail_stress.py 400
echo "== D: + release fence"; COVER_LIB_PATH=$PWD/build_dbg/libcover_tailD.so timeout 900 python tools/dbg/tail_stress.py 400
echo "== C: + release fence + acquire"; COVER_LIB_PATH=$PWD/build_dbg/libcover_tailC.so timeout 900 python tools/dbg/tail_stress.py 400
