import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


# ---- two-rank plumbing run of bench.py (tests/test_multirank_gpu.py) --------------------------------------------------------
# The ranks are FRESH child processes, started here -- after collection, before the first test runs, i.e. before this
# process has made any GPU call (torch.cuda.device_count() does not initialise the GPU on this image) -- and they run while
# the first tests execute. Both ranks share cuda:0 and exchange over gloo: what is tested is the N > 1 code path of bench.py /
# sharding.py on a GPU (prompt sharding, the record all-gather, the winner exchange), not RCCL.
_MULTIRANK = {}


def _launch_ranks(tag, extra, port):
    out = os.path.join(tempfile.mkdtemp(prefix="cover_mr_"), tag)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--small", "--backend", "gloo",
               "--share-gpu", "--no-cpu-baseline", "--no-profile", "--check-out", out] + extra
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    return dict(out=out, procs=procs)


def pytest_collection_finish(session):
    wanted = [it for it in session.items if "test_multirank_gpu.py" in it.nodeid]
    if not wanted:
        return
    import torch
    if torch.cuda.device_count() < 1:
        return
    base = 29700 + (os.getpid() % 1000)
    _MULTIRANK["weak"] = _launch_ranks("weak", ["--scaling", "weak"], base)
    _MULTIRANK["strong"] = _launch_ranks("strong", ["--scaling", "strong"], base + 1)
    _MULTIRANK["config3"] = _launch_ranks("config3", ["--config", "3"], base + 2)


@pytest.fixture(scope="session")
def multirank_runs():
    return _MULTIRANK
