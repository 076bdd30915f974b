"""Serving boundary: wire format against bytes produced by the reference's own msgpack_numpy module
(tests/golden/wire_msgpack.npz, generator: tests/golden/make_wire_golden.py) and the per-connection protocol."""
import os

import numpy as np
import pytest

from cover_vla_amd import server

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wire_msgpack.npz")


def _objects():
    rng = np.random.default_rng(7)   # the generator script's objects, in its order of draws
    return {
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


def _same(a, b):
    if isinstance(a, dict):
        return isinstance(b, dict) and a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return isinstance(b, np.ndarray) and a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
    if isinstance(a, np.generic):
        return isinstance(b, np.generic) and a.dtype == b.dtype and a == b
    return type(a) == type(b) and a == b


def test_wire_format_matches_reference_bytes():
    gold = np.load(GOLD)
    objs = _objects()
    assert set(gold.files) == set(objs)
    for name, obj in objs.items():
        ref_bytes = gold[name].tobytes()
        assert server.pack(obj) == ref_bytes, name                 # encoder: byte for byte what the reference sends
        assert _same(obj, server.unpack(ref_bytes)), name           # decoder: what the reference sends comes back as sent
    back = server.unpack(gold["empty_and_strided"].tobytes())
    assert back["t"].flags["C_CONTIGUOUS"] and back["t"].shape == (4, 3)   # a strided array travels in C order


@pytest.mark.parametrize("bad", [np.array([1 + 2j]), np.array([object()], dtype=object), np.zeros(2, dtype=[("a", "i4")])])
def test_wire_format_refuses_what_the_reference_refuses(bad):
    with pytest.raises(ValueError):
        server.pack({"x": bad})


class _FakePolicy:
    def __init__(self):
        self.calls = []

    def select_action(self, obs):
        self.calls.append(("infer", obs["step"]))
        if obs.get("explode"):
            raise RuntimeError("boom")
        return {"action": np.full((4, 7), float(obs["step"]), dtype=np.float32), "idx": np.int64(obs["step"])}

    def reset(self):
        self.calls.append(("reset",))

    def switch_model(self, path):
        self.calls.append(("switch", path))


def test_session_protocol():
    pol = _FakePolicy()
    s = server.PolicySession(pol, {"policy": "cover", "n_action_steps": 4})
    assert server.unpack(s.greeting()) == {"policy": "cover", "n_action_steps": 4}
    r, close = s.handle(server.pack({"reset": True}))
    assert server.unpack(r) == {"status": "reset"} and not close
    r, close = s.handle(server.pack({"new_model_path": "/ckpt/a"}))
    assert server.unpack(r) == {"status": "model switched"} and not close
    r, close = s.handle(server.pack({"step": 5, "observation.state": np.zeros(7, np.float32)}))
    out = server.unpack(r)
    assert not close and out["idx"] == 5 and np.array_equal(out["action"], np.full((4, 7), 5.0, np.float32))
    r, close = s.handle(server.pack({"step": 6, "explode": True}))      # the traceback text goes out, then the connection closes
    assert close and isinstance(r, str) and "RuntimeError: boom" in r
    r, close = s.handle(b"\\xc1 not msgpack")
    assert close and isinstance(r, str)
    assert pol.calls == [("reset",), ("switch", "/ckpt/a"), ("infer", 5), ("infer", 6)]


def test_verified_policy_composes_sampler_and_verifier():
    seen = {}

    def sample(obs):
        return np.arange(8 * 7, dtype=np.float32).reshape(8, 7), {"groups": 4}

    def choose(cands, ctx, obs):
        seen["ctx"] = ctx
        return cands[obs["pick"]]

    pol = server.VerifiedPolicy(sample, choose)
    s = server.PolicySession(pol)
    out = server.unpack(s.handle(server.pack({"pick": 3}))[0])
    assert np.array_equal(out, np.arange(21, 28, dtype=np.float32)) and seen["ctx"] == {"groups": 4}
    r, close = s.handle(server.pack({"new_model_path": "x"}))           # one checkpoint: refused loudly, connection closed
    assert close and "NotImplementedError" in r


# ------------------------------------------------------------------------------------------------ websocket transport
class _EchoPolicy:
    def __init__(self):
        self.resets, self.paths = 0, []

    def select_action(self, obs):
        if "boom" in obs:
            raise ValueError("scripted failure")
        img = obs["image"]
        return {"actions": (img.astype(np.float32).mean(axis=(0, 1)) + obs["offset"]).astype(np.float32), "n": np.int64(img.size)}

    def reset(self):
        self.resets += 1

    def switch_model(self, path):
        self.paths.append(path)


def test_websocket_round_trip_on_localhost():
    """serve_websocket end to end over a real TCP socket (RFC 6455 transport of cover_vla_amd.wsproto): greeting, an observation
    with a 0.9 MB frame (64-bit length header, masked client frame), reset, model switch, and the error path -- traceback as a TEXT
    frame, then close code 1011 with the reference's reason (websocket_policy_server.py:54-91)."""
    import threading
    from cover_vla_amd import server, wsproto
    pol = _EchoPolicy()
    port_box, ev = [], threading.Event()

    def ready(port):
        port_box.append(port)
        ev.set()

    th = threading.Thread(target=lambda: server.serve_websocket(pol, "127.0.0.1", 0, {"policy": "echo", "horizon": 4}, ready=ready), daemon=True)
    th.start()
    assert ev.wait(20)
    cl = server.WebsocketClientPolicy("127.0.0.1", port_box[0])
    assert cl.metadata == {"policy": "echo", "horizon": 4}
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    out = cl.select_action({"image": img, "offset": np.float32(0.5), "task": "put the spoon on the towel"})
    assert np.allclose(out["actions"], img.astype(np.float32).mean(axis=(0, 1)) + 0.5) and int(out["n"]) == img.size
    assert cl.reset() == {"status": "reset"} and pol.resets == 1
    assert cl.switch_model("/ckpt/a") == {"status": "model switched"} and pol.paths == ["/ckpt/a"]
    small = cl.select_action({"image": img[:4, :4], "offset": np.float32(0.0)})       # 7-bit length frames
    assert small["actions"].shape == (3,)
    with pytest.raises(RuntimeError) as ei:
        cl.select_action({"boom": True})
    assert "scripted failure" in str(ei.value) and "Traceback" in str(ei.value)
    with pytest.raises(wsproto.ConnectionClosed) as ce:                                  # the server closes with INTERNAL_ERROR
        cl._ws.recv()
    assert ce.value.code == server.CLOSE_INTERNAL_ERROR and ce.value.reason == server.CLOSE_REASON
    # a second client on the same server (one session per connection)
    cl2 = server.WebsocketClientPolicy("127.0.0.1", port_box[0])
    assert cl2.reset() == {"status": "reset"} and pol.resets == 2
    cl2.close()


def test_websocket_frame_codec_known_answers():
    from cover_vla_amd import wsproto
    # RFC 6455 section 1.3 handshake example
    assert wsproto.accept_key("dGhlIHNhbXBsZSBub25jZQ==") == "s3pPLMBiTxaQ9kYGzzhZRbK+xOo="
    # RFC 6455 section 5.7: unmasked text "Hello", and the 256-byte binary header 0x82 0x7E 0x0100
    assert wsproto.encode_frame(wsproto.OP_TEXT, b"Hello", False) == bytes([0x81, 0x05, 0x48, 0x65, 0x6C, 0x6C, 0x6F])
    f = wsproto.encode_frame(wsproto.OP_BINARY, bytes(256), False)
    assert f[:4] == bytes([0x82, 0x7E, 0x01, 0x00]) and len(f) == 260
    f = wsproto.encode_frame(wsproto.OP_BINARY, bytes(65536), False)
    assert f[:2] == bytes([0x82, 0x7F]) and f[2:10] == (65536).to_bytes(8, "big")
    m = wsproto.encode_frame(wsproto.OP_TEXT, b"Hello", True)                            # masked: key at [2:6]
    assert m[1] == 0x85 and bytes(b ^ m[2 + (i & 3)] for i, b in enumerate(m[6:])) == b"Hello"


def test_websocket_transport_limits_and_protocol_errors():
    """wsproto server side: max_message_bytes closes with 1009 BEFORE buffering what the peer announces (single frame and fragments),
    an unmasked client frame fails the connection with 1002 (RFC 6455 5.1), an oversized handshake gets a 400, and the default
    (None) stays unlimited like the reference's max_size=None."""
    import asyncio
    import socket
    import struct
    import threading
    from cover_vla_amd import wsproto

    async def echo(conn):
        while True:
            await conn.send(await conn.recv())

    def start(limit):
        box, ev = [], threading.Event()

        def run():
            asyncio.run(wsproto.serve(echo, "127.0.0.1", 0, ready=lambda p: (box.append(p), ev.set()), max_message_bytes=limit))

        threading.Thread(target=run, daemon=True).start()
        assert ev.wait(20)
        return box[0]

    port = start(1000)
    c = wsproto.ClientConnection("127.0.0.1", port)
    c.send(b"x" * 1000)
    assert c.recv() == b"x" * 1000                                  # at the limit: fine
    c.send(b"y" * 1001)
    with pytest.raises(wsproto.ConnectionClosed) as e:
        c.recv()
    assert e.value.code == 1009
    # fragments that add up beyond the limit; the announced 2^40-byte frame is refused from its header alone
    c = wsproto.ClientConnection("127.0.0.1", port)
    key = b"\x01\x02\x03\x04"
    frag = bytes([0x02, 0x80 | 126]) + struct.pack("!H", 600) + key + wsproto._unmask(b"a" * 600, key)     # BINARY, FIN = 0
    cont = bytes([0x80, 0x80 | 126]) + struct.pack("!H", 600) + key + wsproto._unmask(b"b" * 600, key)     # CONT, FIN = 1
    c._s.sendall(frag + cont)
    with pytest.raises(wsproto.ConnectionClosed) as e:
        c.recv()
    assert e.value.code == 1009
    c = wsproto.ClientConnection("127.0.0.1", port)
    c._s.sendall(bytes([0x82, 0x80 | 127]) + struct.pack("!Q", 1 << 40) + key)
    with pytest.raises(wsproto.ConnectionClosed) as e:
        c.recv()
    assert e.value.code == 1009
    # unmasked client frame
    c = wsproto.ClientConnection("127.0.0.1", port)
    c._s.sendall(wsproto.encode_frame(wsproto.OP_BINARY, b"plain", False))
    with pytest.raises(wsproto.ConnectionClosed) as e:
        c.recv()
    assert e.value.code == 1002
    # handshake with a header block beyond the stream limit: 400, no task exception
    s = socket.create_connection(("127.0.0.1", port), timeout=20)
    s.sendall(b"GET / HTTP/1.1\r\nX-Junk: " + b"j" * (80 * 1024) + b"\r\n")
    got = b""
    while b"\r\n\r\n" not in got:
        chunk = s.recv(4096)
        if not chunk:
            break
        got += chunk
    assert got.startswith(b"HTTP/1.1 400")
    s.close()
    # default: unlimited
    port2 = start(None)
    c = wsproto.ClientConnection("127.0.0.1", port2)
    big = os.urandom(300000)
    c.send(big)
    assert c.recv() == big
    c.close()
