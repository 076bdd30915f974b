"""Policy / verifier serving boundary: the wire format and the per-connection protocol of the reference's
policy server, transport-agnostic.

Reference behaviour restated here (not its code):
  * wire format  INT-ACT/packages/policy-server-client/src/policy_server_client/msgpack_numpy.py:21-57 -- msgpack with two
    extension dicts, keys as BYTES: an ndarray travels as {b"__ndarray__": True, b"data": raw bytes (C order), b"dtype":
    numpy dtype string, b"shape": tuple}, a numpy scalar as {b"__npgeneric__": True, b"data": python value, b"dtype": str};
    void / object / complex dtypes are refused (ValueError);
  * protocol     websocket_policy_server.py:54-91 -- on connect the server sends the packed metadata dict; every request is
    one packed dict: {"new_model_path": p} -> policy.switch_model(p), reply {"status": "model switched"};
    {"reset": True} -> policy.reset(), reply {"status": "reset"}; anything else is an observation ->
    policy.select_action(obs), reply = the packed action. On an exception the traceback TEXT is sent and the connection is
    closed with the websocket INTERNAL_ERROR code (1011) and the reason string below.
`PolicySession` turns one received message into one reply and says whether to close (transport-agnostic); `serve_websocket`
runs it over the `websockets` package when installed, else over the RFC 6455 transport of cover_vla_amd.wsproto.
"""
import traceback
from typing import Any, Callable, Optional, Tuple, Union

import msgpack
import numpy as np

CLOSE_INTERNAL_ERROR = 1011
CLOSE_REASON = "Internal server error. Traceback included in previous frame."
_REFUSED_KINDS = frozenset("VOc")


def _to_wire(obj: Any) -> Any:
    """msgpack `default` hook: numpy values -> the reference's tagged dicts; everything else is not ours to encode."""
    if isinstance(obj, (np.ndarray, np.generic)):
        if obj.dtype.kind in _REFUSED_KINDS:
            raise ValueError(f"Unsupported dtype: {obj.dtype}")
        if isinstance(obj, np.ndarray):
            return {b"__ndarray__": True, b"data": obj.tobytes(), b"dtype": obj.dtype.str, b"shape": obj.shape}
        return {b"__npgeneric__": True, b"data": obj.item(), b"dtype": obj.dtype.str}
    return obj


def _from_wire(d: dict) -> Any:
    """msgpack `object_hook`: tagged dicts -> numpy values (arrays are read-only views of the received buffer, as in the reference)."""
    if b"__ndarray__" in d:
        return np.ndarray(shape=d[b"shape"], dtype=np.dtype(d[b"dtype"]), buffer=d[b"data"])
    if b"__npgeneric__" in d:
        return np.dtype(d[b"dtype"]).type(d[b"data"])
    return d


def pack(obj: Any) -> bytes:
    return msgpack.packb(obj, default=_to_wire)


def unpack(data: Union[bytes, bytearray, memoryview]) -> Any:
    return msgpack.unpackb(data, object_hook=_from_wire)


class PolicySession:
    """One client connection. `policy` needs select_action(obs) and may have reset() / switch_model(path)."""

    def __init__(self, policy: Any, metadata: Optional[dict] = None):
        self.policy = policy
        self.metadata = dict(metadata or {})
        self._packer = msgpack.Packer(default=_to_wire)

    def greeting(self) -> bytes:
        return self._packer.pack(self.metadata)

    def handle(self, message: Union[bytes, bytearray, memoryview]) -> Tuple[Union[bytes, str], bool]:
        """-> (reply frame, close connection?). A str reply is the traceback of a failed request (text frame, then close)."""
        try:
            obs = unpack(message)
            path = obs.get("new_model_path", None)
            if path is not None:
                self.policy.switch_model(path)
                return self._packer.pack({"status": "model switched"}), False
            if obs.get("reset", False):
                self.policy.reset()
                return self._packer.pack({"status": "reset"}), False
            return self._packer.pack(self.policy.select_action(obs)), False
        except Exception:
            return traceback.format_exc(), True


class VerifiedPolicy:
    """select_action(obs) = sample candidates with `sampler`, score them with `verifier`, return the chosen action chunk:
    the object the session serves when the MI355X node does both halves of the decision.
    `sample(obs) -> (candidates, context)` and `choose(candidates, context, obs) -> action` are injected callables (the
    eval driver's batch construction and `host.verify_and_select` respectively)."""

    def __init__(self, sample: Callable[[dict], Tuple[Any, Any]], choose: Callable[[Any, Any, dict], Any],
                 reset: Optional[Callable[[], None]] = None, switch_model: Optional[Callable[[str], None]] = None):
        self._sample, self._choose, self._reset, self._switch = sample, choose, reset, switch_model

    def select_action(self, obs: dict):
        candidates, ctx = self._sample(obs)
        return self._choose(candidates, ctx, obs)

    def reset(self):
        if self._reset is not None:
            self._reset()

    def switch_model(self, path: str):
        if self._switch is None:
            raise NotImplementedError("this server holds one checkpoint")
        self._switch(path)


async def _serve(conn_recv_send_close, session: PolicySession) -> None:
    """The per-connection loop of websocket_policy_server.py:54-91 over any connection object with send / recv / close."""
    ws = conn_recv_send_close
    await ws.send(session.greeting())
    while True:
        msg = await ws.recv()                       # raises the transport's ConnectionClosed when the client leaves
        reply, close = session.handle(msg)
        await ws.send(reply)
        if close:
            await ws.close(code=CLOSE_INTERNAL_ERROR, reason=CLOSE_REASON)
            return


def serve_websocket(policy: Any, host: str = "0.0.0.0", port: int = 8000, metadata: Optional[dict] = None,
                    ready: Optional[Callable[[int], None]] = None, max_message_bytes: Optional[int] = None) -> None:
    """Blocking websocket server with the reference's framing (binary msgpack frames, no compression, no size limit unless
    `max_message_bytes` is given -- the reference passes max_size=None, websocket_policy_server.py:44-49; traceback as a text
    frame + close code 1011 on error). Runs on the `websockets` package when it is installed (the reference's
    transport), else on the RFC 6455 transport in cover_vla_amd.wsproto (standard library only). `ready(port)` is called once the
    socket listens (port 0 = pick a free one)."""
    import asyncio
    try:
        import websockets
        import websockets.asyncio.server
    except ImportError:
        websockets = None

    if websockets is not None:   # pragma: no cover - depends on the deployment image
        async def handler(ws):
            try:
                await _serve(ws, PolicySession(policy, metadata))
            except websockets.ConnectionClosed:
                return

        async def run():
            async with websockets.asyncio.server.serve(handler, host, port, compression=None, max_size=max_message_bytes) as server:
                if ready is not None:
                    ready(server.sockets[0].getsockname()[1])   # the BOUND port (port = 0 picks a free one)
                await server.serve_forever()
    else:
        from . import wsproto

        async def handler(ws):
            try:
                await _serve(ws, PolicySession(policy, metadata))
            except wsproto.ConnectionClosed:
                return

        async def run():
            await wsproto.serve(handler, host, port, ready=ready, max_message_bytes=max_message_bytes)

    asyncio.run(run())


class WebsocketClientPolicy:
    """Client side of the same protocol (the simulator's wrapper, websocket_client_policy.py in the reference's package):
    connect, read the metadata greeting, then select_action(obs) / reset() / switch_model(path) as request-reply pairs."""

    def __init__(self, host: str = "127.0.0.1", port: int = 8000, timeout: float = 60.0):
        from . import wsproto
        self._ws = wsproto.ClientConnection(host, port, timeout)
        self.metadata = unpack(self._ws.recv())

    def _call(self, obj):
        self._ws.send(pack(obj))
        reply = self._ws.recv()
        if isinstance(reply, str):
            raise RuntimeError(f"Error in policy server:\n{reply}")
        return unpack(reply)

    def select_action(self, obs: dict):
        return self._call(obs)

    def reset(self):
        return self._call({"reset": True})

    def switch_model(self, path: str):
        return self._call({"new_model_path": path})

    def close(self):
        self._ws.close()
