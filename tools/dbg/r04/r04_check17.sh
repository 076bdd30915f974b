set -x
cd $GRAFT_REPO_ROOT
COVER_TAIL_REDUCE=1 timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -x -k "config2_n16" 2>&1 | tail -3
COVER_TAIL_REDUCE=1 COVER_TAIL_MEMSET=1 timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -x -k "config2_n16" 2>&1 | tail -3
COVER_TAIL_REDUCE=1 timeout 900 python -m pytest tests/test_openvla_gpu.py -q -x -k "graph_replay" 2>&1 | tail -3
