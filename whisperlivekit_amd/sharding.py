"""One stream per GPU: stream -> rank assignment and the single collective of this path, the
start-up broadcast of the packed weight arena from rank 0 over RCCL/xGMI.

Sessions never talk to each other (a session's state is private and the model is read-only), so
there is no steady-state collective: each rank runs its own streams against its own weight replica."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .dims import ModelDims
from .engine import HipWhisperModel, _cdims, arena_floats, packed_tensor_names


def assign_streams(n_streams: int, world_size: int) -> List[List[int]]:
    """stream i -> rank i mod world_size."""
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in range(n_streams):
        out[i % world_size].append(i)
    return out


def pack_arena_host(dims: ModelDims, packed: Mapping[str, np.ndarray]) -> np.ndarray:
    """Lay the packed tensors out in one flat fp32 array exactly as the device arena is laid out
    (offsets from wlk_tensor_lookup); needs no GPU."""
    lib = _lib.load()
    cd = _cdims(dims)
    flat = np.zeros(arena_floats(dims), np.float32)
    for name in packed_tensor_names(dims):
        off, numel = C.c_uint64(), C.c_uint64()
        _lib.check(lib.wlk_tensor_lookup(C.byref(cd), name.encode(), C.byref(off), C.byref(numel)))
        a = np.asarray(packed[name], np.float32).reshape(-1)
        if a.size != numel.value:
            raise ValueError(f"{name}: expected {numel.value} values, got {a.size}")
        flat[off.value: off.value + numel.value] = a
    return flat


def dist_env() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; (0, 1, 0) when absent."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)),
            int(os.environ.get("LOCAL_RANK", 0)))


def init_process_group(backend: Optional[str] = None):
    """backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests.  Rendezvous comes from MASTER_ADDR /
    MASTER_PORT (use 127.0.0.1 on a single node)."""
    import torch
    import torch.distributed as dist
    rank, world, local = dist_env()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_arena(arena, src: int = 0):
    """In-place broadcast of the weight arena tensor (CUDA tensor -> RCCL, CPU tensor -> gloo)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(arena, src=src)
    return arena


def replicated_model(dims: ModelDims, packed: Optional[Mapping[str, np.ndarray]],
                     alignment_heads: Sequence[Tuple[int, int]], device: int, src: int = 0,
                     timing: Optional[dict] = None) -> HipWhisperModel:
    """Every rank gets a full weight replica: rank ``src`` packs the arena on the host, all ranks
    allocate it as a torch CUDA tensor, one RCCL broadcast fills it, the library adopts the pointer.
    ``timing`` (optional dict) receives ``broadcast_ms``: wall time of the collective on this rank, and ``finalize_ms``:
    the rank's own wlk_model_finalize (the X3 weight images are packed per rank from the broadcast fp32 arena - 1.5 x its
    bytes stay local instead of crossing xGMI)."""
    import time
    import torch
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = arena_floats(dims)
    if rank == src:
        if packed is None:
            raise ValueError("the source rank needs the packed weights")
        arena = torch.from_numpy(pack_arena_host(dims, packed)).to(f"cuda:{device}")
    else:
        arena = torch.empty(n, dtype=torch.float32, device=f"cuda:{device}")
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    broadcast_arena(arena, src)
    torch.cuda.synchronize(device)
    if timing is not None:
        timing["broadcast_ms"] = round(1e3 * (time.perf_counter() - t0), 3)
        timing["arena_bytes"] = int(n) * 4
    model = HipWhisperModel(dims, device, arena=arena)
    model.set_alignment_heads(alignment_heads)
    t1 = time.perf_counter()
    model.finalize()          # per rank, after the broadcast: derived weight images (X3 planes, the stacked cross-K|V matrix)
    if timing is not None:
        timing["finalize_ms"] = round(1e3 * (time.perf_counter() - t1), 3)
    return model
