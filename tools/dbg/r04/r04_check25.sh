set -x
cd $GRAFT_REPO_ROOT
echo "== no DMA, no barrier"; COVER_LIB_PATH=$PWD/build_dbg/libcover_dbgabl28.so M=448 python tools/dbg/exp_pc_debug.py 2>&1 | tail -3
echo "== no DMA, no barrier, no LDS reads (MFMA only)"; COVER_LIB_PATH=$PWD/build_dbg/libcover_dbgabl30.so M=448 python tools/dbg/exp_pc_debug.py 2>&1 | tail -3
