set -x
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do COVER_TAIL_REDUCE=1 timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -x -k "config2_n16" 2>&1 | tail -3; done
COVER_TAIL_REDUCE=1 COVER_DECODE_GRAPH=0 timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -x -k "config2_n16" 2>&1 | tail -3
COVER_TAIL_REDUCE=0 timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -x -k "config2_n16" 2>&1 | tail -3
