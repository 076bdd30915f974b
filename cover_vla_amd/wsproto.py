"""Minimal RFC 6455 WebSocket transport (server + blocking client) on asyncio / sockets from the standard library.

The reference serves its policy with the `websockets` package
(INT-ACT/packages/policy-server-client/src/policy_server_client/websocket_policy_server.py:40-91: `serve(handler, host, port,
compression=None, max_size=None)`, binary msgpack frames, a TEXT frame with the traceback and close code 1011 on error). That
package is not in this image, so the few pieces of the protocol the path uses are restated here from the RFC: the HTTP/1.1
Upgrade handshake (Sec-WebSocket-Accept = base64(sha1(key + GUID))), unfragmented and fragmented data frames with 7 / 16 /
64-bit lengths, client-to-server masking, ping/pong and the close handshake. No extensions (the reference disables
compression). Message size: unlimited by default like the reference's `max_size=None`; `max_message_bytes` (serve / ServerConnection)
closes with 1009 before buffering a larger message. An unmasked client frame fails the connection with 1002 (RFC 6455 5.1).
`cover_vla_amd.server.serve_websocket` runs on this when `websockets` is absent; an unmodified `websockets` client (the
simulator side) talks to it.
"""
from __future__ import annotations

import asyncio
import base64
import hashlib
import os
import socket
import struct
from typing import Awaitable, Callable, Optional, Tuple, Union

GUID = "258EAFA5-E914-47DA-95CA-C5AB0DC85B11"
OP_CONT, OP_TEXT, OP_BINARY, OP_CLOSE, OP_PING, OP_PONG = 0x0, 0x1, 0x2, 0x8, 0x9, 0xA


class ConnectionClosed(Exception):
    def __init__(self, code: int = 1005, reason: str = ""):
        super().__init__(f"websocket closed: {code} {reason}")
        self.code, self.reason = code, reason


def accept_key(key: str) -> str:
    return base64.b64encode(hashlib.sha1((key + GUID).encode()).digest()).decode()


def encode_frame(opcode: int, payload: bytes, mask: bool) -> bytes:
    n = len(payload)
    head = bytes([0x80 | opcode])
    mbit = 0x80 if mask else 0
    if n < 126:
        head += bytes([mbit | n])
    elif n < (1 << 16):
        head += bytes([mbit | 126]) + struct.pack("!H", n)
    else:
        head += bytes([mbit | 127]) + struct.pack("!Q", n)
    if not mask:
        return head + payload
    key = os.urandom(4)
    masked = bytes(b ^ key[i & 3] for i, b in enumerate(payload)) if n < 4096 else _mask_fast(payload, key)
    return head + key + masked


def _mask_fast(payload: bytes, key: bytes) -> bytes:
    import numpy as np
    a = np.frombuffer(payload, dtype=np.uint8)
    k = np.frombuffer((key * (len(payload) // 4 + 1))[: len(payload)], dtype=np.uint8)
    return (a ^ k).tobytes()


def _unmask(payload: bytes, key: bytes) -> bytes:
    return _mask_fast(payload, key) if len(payload) >= 4096 else bytes(b ^ key[i & 3] for i, b in enumerate(payload))


def decode_header(b0: int, b1: int) -> Tuple[bool, int, bool, int]:
    return bool(b0 & 0x80), b0 & 0x0F, bool(b1 & 0x80), b1 & 0x7F


# ------------------------------------------------------------------------------------------------ server (asyncio)
class ServerConnection:
    """One accepted connection: `await recv()` -> bytes (binary) or str (text), `await send(bytes | str)`, `await close(code, reason)`."""

    def __init__(self, reader: asyncio.StreamReader, writer: asyncio.StreamWriter, max_message_bytes: Optional[int] = None):
        self._r, self._w, self._closed = reader, writer, False
        self._max = max_message_bytes
        self.remote_address = writer.get_extra_info("peername")

    async def _fail(self, code: int, reason: str):
        await self.close(code, reason)
        raise ConnectionClosed(code, reason)

    async def _read_frame(self, have: int):
        h = await self._r.readexactly(2)
        fin, op, masked, n = decode_header(h[0], h[1])
        if n == 126:
            n = struct.unpack("!H", await self._r.readexactly(2))[0]
        elif n == 127:
            n = struct.unpack("!Q", await self._r.readexactly(8))[0]
        if not masked:                                   # RFC 6455 5.1: a server MUST close on an unmasked client frame
            await self._fail(1002, "unmasked client frame")
        if op >= 0x8 and (n > 125 or not fin):           # 5.5: control frames are short and never fragmented
            await self._fail(1002, "malformed control frame")
        if self._max is not None and have + n > self._max:   # refuse BEFORE buffering what the peer announces
            await self._fail(1009, "message too big")
        key = await self._r.readexactly(4)
        payload = await self._r.readexactly(n) if n else b""
        return fin, op, _unmask(payload, key)

    async def recv(self) -> Union[bytes, str]:
        parts, kind, have = [], None, 0
        while True:
            try:
                fin, op, payload = await self._read_frame(have)
            except (asyncio.IncompleteReadError, ConnectionError):
                self._closed = True
                raise ConnectionClosed(1006, "connection lost")
            if op == OP_PING:
                self._w.write(encode_frame(OP_PONG, payload, False))
                await self._w.drain()
                continue
            if op == OP_PONG:
                continue
            if op == OP_CLOSE:
                code = struct.unpack("!H", payload[:2])[0] if len(payload) >= 2 else 1005
                if not self._closed:
                    self._w.write(encode_frame(OP_CLOSE, payload[:2], False))
                    await self._w.drain()
                    self._closed = True
                self._w.close()
                raise ConnectionClosed(code, payload[2:].decode(errors="replace"))
            if op in (OP_TEXT, OP_BINARY):
                kind = op
            parts.append(payload)
            have += len(payload)
            if fin:
                data = b"".join(parts)
                return data.decode() if kind == OP_TEXT else data

    async def send(self, data: Union[bytes, bytearray, memoryview, str]) -> None:
        if isinstance(data, str):
            self._w.write(encode_frame(OP_TEXT, data.encode(), False))
        else:
            self._w.write(encode_frame(OP_BINARY, bytes(data), False))
        await self._w.drain()

    async def close(self, code: int = 1000, reason: str = "") -> None:
        if not self._closed:
            self._closed = True
            self._w.write(encode_frame(OP_CLOSE, struct.pack("!H", code) + reason.encode(), False))
            await self._w.drain()
        try:
            self._w.close()
        except Exception:
            pass


async def _handshake_server(reader: asyncio.StreamReader, writer: asyncio.StreamWriter) -> bool:
    try:
        request = await reader.readuntil(b"\r\n\r\n")
    except (asyncio.LimitOverrunError, ValueError):      # a header block beyond the stream limit (64 KiB)
        writer.write(b"HTTP/1.1 400 Bad Request\r\nConnection: close\r\n\r\n")
        await writer.drain()
        writer.close()
        return False
    lines = request.decode(errors="replace").split("\r\n")
    headers = {}
    for ln in lines[1:]:
        if ":" in ln:
            k, v = ln.split(":", 1)
            headers[k.strip().lower()] = v.strip()
    key = headers.get("sec-websocket-key")
    if not lines[0].startswith("GET") or key is None or "websocket" not in headers.get("upgrade", "").lower():
        writer.write(b"HTTP/1.1 400 Bad Request\r\nConnection: close\r\n\r\n")
        await writer.drain()
        writer.close()
        return False
    writer.write(("HTTP/1.1 101 Switching Protocols\r\nUpgrade: websocket\r\nConnection: Upgrade\r\n"
                  f"Sec-WebSocket-Accept: {accept_key(key)}\r\n\r\n").encode())
    await writer.drain()
    return True


async def serve(handler: Callable[[ServerConnection], Awaitable[None]], host: str, port: int, ready: Optional[Callable[[int], None]] = None,
                max_message_bytes: Optional[int] = None):
    """Run `handler(connection)` for every client until cancelled. `ready(port)` is called once the socket listens.
    max_message_bytes: None = unlimited (the reference's max_size=None); otherwise larger messages close the connection with 1009."""
    async def on_client(reader, writer):
        try:
            if await _handshake_server(reader, writer):
                await handler(ServerConnection(reader, writer, max_message_bytes))
        except (ConnectionClosed, asyncio.IncompleteReadError, ConnectionError):
            pass
        finally:
            try:
                writer.close()
            except Exception:
                pass

    server = await asyncio.start_server(on_client, host, port)
    if ready is not None:
        ready(server.sockets[0].getsockname()[1])
    async with server:
        await server.serve_forever()


# ------------------------------------------------------------------------------------------------ client (blocking)
class ClientConnection:
    """Blocking client (what the simulator-side wrapper needs: connect, recv greeting, send obs, recv action)."""

    def __init__(self, host: str, port: int, timeout: float = 30.0):
        self._s = socket.create_connection((host, port), timeout=timeout)
        key = base64.b64encode(os.urandom(16)).decode()
        self._s.sendall((f"GET / HTTP/1.1\r\nHost: {host}:{port}\r\nUpgrade: websocket\r\nConnection: Upgrade\r\n"
                         f"Sec-WebSocket-Key: {key}\r\nSec-WebSocket-Version: 13\r\n\r\n").encode())
        resp = b""
        while b"\r\n\r\n" not in resp:
            chunk = self._s.recv(4096)
            if not chunk:
                raise ConnectionError("handshake: connection closed")
            resp += chunk
        head, self._buf = resp.split(b"\r\n\r\n", 1)
        if b" 101 " not in head.split(b"\r\n")[0] or accept_key(key).encode() not in head:
            raise ConnectionError(f"handshake refused: {head[:200]!r}")

    def _read(self, n: int) -> bytes:
        while len(self._buf) < n:
            chunk = self._s.recv(max(65536, n - len(self._buf)))
            if not chunk:
                raise ConnectionClosed(1006, "connection lost")
            self._buf += chunk
        out, self._buf = self._buf[:n], self._buf[n:]
        return out

    def recv(self) -> Union[bytes, str]:
        parts, kind = [], None
        while True:
            h = self._read(2)
            fin, op, masked, n = decode_header(h[0], h[1])
            if n == 126:
                n = struct.unpack("!H", self._read(2))[0]
            elif n == 127:
                n = struct.unpack("!Q", self._read(8))[0]
            key = self._read(4) if masked else None
            payload = self._read(n) if n else b""
            if key is not None:
                payload = _unmask(payload, key)
            if op == OP_PING:
                self._s.sendall(encode_frame(OP_PONG, payload, True))
                continue
            if op == OP_PONG:
                continue
            if op == OP_CLOSE:
                code = struct.unpack("!H", payload[:2])[0] if len(payload) >= 2 else 1005
                try:
                    self._s.sendall(encode_frame(OP_CLOSE, payload[:2], True))
                except OSError:
                    pass
                self._s.close()
                raise ConnectionClosed(code, payload[2:].decode(errors="replace"))
            if op in (OP_TEXT, OP_BINARY):
                kind = op
            parts.append(payload)
            if fin:
                data = b"".join(parts)
                return data.decode() if kind == OP_TEXT else data

    def send(self, data: Union[bytes, str]) -> None:
        self._s.sendall(encode_frame(OP_TEXT, data.encode(), True) if isinstance(data, str) else encode_frame(OP_BINARY, bytes(data), True))

    def close(self) -> None:
        try:
            self._s.sendall(encode_frame(OP_CLOSE, struct.pack("!H", 1000), True))
        except OSError:
            pass
        self._s.close()
