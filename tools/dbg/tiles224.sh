export M=448
for p in f h; do COLD=1 COVER_TILE_PICK=$p python tools/dbg/exp_tiles.py 2>&1 | tail -1 | sed "s/^/cold /"; done
for p in auto f; do COLD=1 LDA_PAD=64 COVER_TILE_PICK=$p python tools/dbg/exp_tiles.py 2>&1 | tail -1 | sed "s/^/cold pad64 /"; done
for p in auto f g; do COLD=0 COVER_TILE_PICK=$p python tools/dbg/exp_tiles.py 2>&1 | tail -1 | sed "s/^/warm /"; done
