set -x
cd $GRAFT_REPO_ROOT
echo "== A: as shipped"; timeout 1500 python tools/dbg/tail_stress.py 8000
echo "== C: + release fence + acquire"; COVER_LIB_PATH=$PWD/build_dbg/libcover_tailC.so timeout 1500 python tools/dbg/tail_stress.py 8000
