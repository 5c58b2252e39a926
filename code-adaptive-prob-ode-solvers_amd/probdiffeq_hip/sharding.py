"""Multi-GPU sharding of a batch of IVPs: one process per GPU, no data-path collective.

The path shards trivially: trajectories are independent, so each rank integrates its
own slice with its own host controller. The only exchange is the final gather of per-trajectory
results: ONE `ncclAllGather` over RCCL (xGMI), bound here with ctypes on `librccl.so` and
`libamdhip64.so` -- no torch on this path. The 128-byte `ncclUniqueId` of rank 0 reaches the other
ranks over a plain TCP socket on MASTER_ADDR (MASTER_PORT + 29 by default), or through any
`exchange_id` callable the caller supplies (bench.py hands in the process group the driver's launch
contract already created).

Reference: the reference has no multi-process code at all ("NCCL/collective call
sites: none"); BASELINE.json's north star defines this gather.
"""

from __future__ import annotations

import ctypes as C
import os
import socket
import time

import numpy as np


def shard_bounds(total: int, world: int, rank: int):
    """Contiguous slice [lo, hi) of `total` trajectories owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def interleaved_indices(total: int, world: int, rank: int):
    """Round-robin assignment (rank, rank + world, ...): balances tolerance sweeps, whose step counts
    grow monotonically along the batch."""
    return np.arange(rank, total, world)


# ------------------------------------------------------------------------------------------------
# RCCL through ctypes
# ------------------------------------------------------------------------------------------------
NCCL_UNIQUE_ID_BYTES = 128
_NCCL_CHAR, _NCCL_FLOAT64 = 0, 8  # ncclDataType_t: ncclInt8/ncclChar = 0, ncclFloat64/ncclDouble = 8


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * NCCL_UNIQUE_ID_BYTES)]


def _load(names):
    last = None
    for name in names:
        try:
            return C.CDLL(name)
        except OSError as e:  # pragma: no cover - depends on the box
            last = e
    raise OSError(f"none of {names} could be loaded: {last}")


_libs = None


def _rccl_and_hip():
    global _libs
    if _libs is None:
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        rccl = _load(["librccl.so", "librccl.so.1", os.path.join(rocm, "lib", "librccl.so")])
        hip = _load(["libamdhip64.so", os.path.join(rocm, "lib", "libamdhip64.so")])
        vp = C.c_void_p
        rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        rccl.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, _UniqueId, C.c_int]
        rccl.ncclAllGather.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp]
        rccl.ncclCommDestroy.argtypes = [vp]
        rccl.ncclGetErrorString.argtypes = [C.c_int]
        rccl.ncclGetErrorString.restype = C.c_char_p
        hip.hipSetDevice.argtypes = [C.c_int]
        hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
        hip.hipFree.argtypes = [vp]
        hip.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
        hip.hipStreamSynchronize.argtypes = [vp]
        hip.hipGetErrorString.argtypes = [C.c_int]
        hip.hipGetErrorString.restype = C.c_char_p
        _libs = (rccl, hip)
    return _libs


def _tcp_exchange(payload: bytes | None, rank: int, world: int, addr: str, port: int, timeout: float = 120.0) -> bytes:
    """Rank 0 serves `payload` to the world - 1 other ranks; they connect (with retries) and read it."""
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        srv.settimeout(timeout)
        try:
            for _ in range(world - 1):
                conn, _ = srv.accept()
                with conn:
                    conn.sendall(payload)
        finally:
            srv.close()
        return payload
    deadline = time.monotonic() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as s:
                buf = b""
                while len(buf) < NCCL_UNIQUE_ID_BYTES:
                    chunk = s.recv(NCCL_UNIQUE_ID_BYTES - len(buf))
                    if not chunk:
                        break
                    buf += chunk
                if len(buf) == NCCL_UNIQUE_ID_BYTES:
                    return buf
        except OSError:
            pass
        if time.monotonic() > deadline:
            raise TimeoutError("no ncclUniqueId from rank 0")
        time.sleep(0.05)


class RcclGather:
    """One RCCL communicator for the result gather: `RcclGather(rank, world, device).all_gather(array)`."""

    def __init__(self, rank: int, world: int, device: int, exchange_id=None):
        self.rank, self.world, self.device = rank, world, device
        rccl, hip = _rccl_and_hip()
        self._rccl, self._hip = rccl, hip
        self._hipck(hip.hipSetDevice(device))
        uid = _UniqueId()
        if rank == 0:
            self._ncclck(rccl.ncclGetUniqueId(C.byref(uid)))
        raw = bytes(bytearray(uid.internal)) if rank == 0 else None
        if world > 1:
            if exchange_id is None:
                addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
                port = int(os.environ.get("PNODE_RCCL_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 29))
                raw = _tcp_exchange(raw, rank, world, addr, port)
            else:
                raw = exchange_id(raw)
            C.memmove(C.byref(uid), raw, NCCL_UNIQUE_ID_BYTES)
        self._comm = C.c_void_p()
        self._ncclck(rccl.ncclCommInitRank(C.byref(self._comm), world, uid, rank))

    def _hipck(self, rc):
        if rc != 0:
            raise RuntimeError(f"HIP error {rc}: {self._hip.hipGetErrorString(rc).decode()}")

    def _ncclck(self, rc):
        if rc != 0:
            raise RuntimeError(f"RCCL error {rc}: {self._rccl.ncclGetErrorString(rc).decode()}")

    def all_gather(self, local: np.ndarray) -> np.ndarray:
        """Equal-shaped per-rank arrays, concatenated along axis 0 on every rank (device buffers, one ncclAllGather)."""
        local = np.ascontiguousarray(local)
        nbytes = local.nbytes
        hip, rccl = self._hip, self._rccl
        self._hipck(hip.hipSetDevice(self.device))
        send, recv = C.c_void_p(), C.c_void_p()
        self._hipck(hip.hipMalloc(C.byref(send), max(nbytes, 1)))
        try:
            self._hipck(hip.hipMalloc(C.byref(recv), max(nbytes * self.world, 1)))
            try:
                self._hipck(hip.hipMemcpy(send, local.ctypes.data_as(C.c_void_p), nbytes, 1))  # hipMemcpyHostToDevice
                if local.dtype == np.float64:
                    self._ncclck(rccl.ncclAllGather(send, recv, local.size, _NCCL_FLOAT64, self._comm, None))
                else:
                    self._ncclck(rccl.ncclAllGather(send, recv, nbytes, _NCCL_CHAR, self._comm, None))
                self._hipck(hip.hipStreamSynchronize(None))
                out = np.empty((self.world * local.shape[0],) + local.shape[1:], dtype=local.dtype)
                self._hipck(hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), recv, nbytes * self.world, 2))  # DeviceToHost
            finally:
                hip.hipFree(recv)
        finally:
            hip.hipFree(send)
        return out

    def close(self):
        if self._comm:
            self._rccl.ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()


def gather_results(local: np.ndarray, comm=None) -> np.ndarray:
    """All-gather equally-shaped per-rank result arrays along axis 0 through `comm` (an `RcclGather`, or any object with
    `all_gather(ndarray) -> ndarray`, e.g. the gloo stand-in of the CPU test); identity without a communicator."""
    if comm is None:
        return local
    return comm.all_gather(local)


class GlooGather:
    """CPU stand-in with the same interface, over an initialised torch.distributed (gloo) group: used by the
    world-size-2 CPU test only (RCCL needs GPUs)."""

    def __init__(self):
        import torch.distributed as dist

        self._dist = dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def all_gather(self, local: np.ndarray) -> np.ndarray:
        import torch

        t = torch.from_numpy(np.ascontiguousarray(local))
        out = [torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(out, t)
        return torch.cat(out, dim=0).numpy()

    def close(self):
        pass
