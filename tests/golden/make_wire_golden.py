"""Generates tests/golden/wire_msgpack.npz by running the REFERENCE's own msgpack_numpy module (imported from /root/reference,
which exists only in the build container) on a fixed set of objects. Run once: python tests/golden/make_wire_golden.py"""
import importlib.util, os, sys
import numpy as np

REF = "/root/reference/INT-ACT/packages/policy-server-client/src/policy_server_client/msgpack_numpy.py"
spec = importlib.util.spec_from_file_location("ref_msgpack_numpy", REF)
ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)

rng = np.random.default_rng(7)
objects = {
    "observation": {"observation.images.top": rng.integers(0, 256, (6, 8, 3), dtype=np.uint8),
                    "observation.state": rng.standard_normal(7).astype(np.float32),
                    "task": "put the spoon on the towel", "step": 3},
    "action_chunk": rng.standard_normal((4, 7)).astype(np.float64),
    "scalars": {"score": np.float32(0.125), "idx": np.int64(17), "flag": np.bool_(True), "plain": [1, 2.5, None, "x", b"raw"]},
    "reset": {"reset": True},
    "switch": {"new_model_path": "/ckpt/step_20000"},
    "status": {"status": "model switched"},
    "empty_and_strided": {"e": np.zeros((0, 7), dtype=np.float32), "t": np.arange(12, dtype=np.int16).reshape(3, 4).T},
}
out = {k: np.frombuffer(ref.packb(v), dtype=np.uint8) for k, v in objects.items()}
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "wire_msgpack.npz"), **out)
print({k: len(v) for k, v in out.items()})
