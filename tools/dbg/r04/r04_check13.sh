set -x
cd $GRAFT_REPO_ROOT
COVER_TAIL_REDUCE=1 timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_chain_gpu.py 2>&1 | tail -6
COVER_DECODE_CHAIN=1 timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_openvla_gpu.py tests/test_multirank_gpu.py -m gpu -q 2>&1 | tail -6
python -c "
import torch
from cover_vla_amd import ops
torch.cuda.synchronize(); ops.gemm_tail_status(); ops.decode_chain_status(); print('status ok')"
