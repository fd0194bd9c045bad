"""The N>1 path on CPU: two processes, gloo backend.  Rank 0 packs the weight arena on the host, the
single collective of the path (broadcast of the arena) delivers it, rank 1 checks it bit for bit against
its own packing of the same seeded checkpoint; stream -> rank assignment is the documented i mod G."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from whisperlivekit_amd import sharding, synth
from whisperlivekit_amd.dims import MODEL_DIMS
from whisperlivekit_amd.engine import arena_floats, pack_state_dict


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ok):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = sharding.init_process_group("gloo")
    assert (r, w) == (rank, world)
    dims = MODEL_DIMS["micro.en"]
    n = arena_floats(dims)
    if rank == 0:
        arena = torch.from_numpy(sharding.pack_arena_host(dims, pack_state_dict(dims, synth.synth_state_dict(dims, 0))))
    else:
        arena = torch.full((n,), float("nan"))
    sharding.broadcast_arena(arena, src=0)
    mine = sharding.pack_arena_host(dims, pack_state_dict(dims, synth.synth_state_dict(dims, 0)))
    same = bool(np.array_equal(arena.numpy(), mine))
    # every rank must agree on the stream sharding
    mine_streams = sharding.assign_streams(8, world)[rank]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine_streams)
    flat = sorted(x for g in gathered for x in g)
    ok[rank] = int(same and flat == list(range(8)) and all(s % world == rank for s in mine_streams))
    dist.destroy_process_group()


def test_arena_broadcast_and_stream_sharding_world2():
    world = 2
    port = _free_port()
    ok = mp.get_context("spawn").Array("i", [0] * world)
    mp.spawn(_worker, args=(world, port, ok), nprocs=world, join=True)
    assert list(ok) == [1] * world


def test_assign_streams():
    assert sharding.assign_streams(8, 1) == [[0, 1, 2, 3, 4, 5, 6, 7]]
    assert sharding.assign_streams(8, 8) == [[i] for i in range(8)]
    assert sharding.assign_streams(8, 4) == [[0, 4], [1, 5], [2, 6], [3, 7]]
    assert sharding.assign_streams(3, 4) == [[0], [1], [2], []]


def test_pack_arena_host_places_every_tensor():
    dims = MODEL_DIMS["micro.en"]
    packed = pack_state_dict(dims, synth.synth_state_dict(dims, 1))
    flat = sharding.pack_arena_host(dims, packed)
    assert flat.size == arena_floats(dims)
    import ctypes as C
    from whisperlivekit_amd import _lib
    from whisperlivekit_amd.engine import _cdims
    off, numel = C.c_uint64(), C.c_uint64()
    _lib.check(_lib.load().wlk_tensor_lookup(C.byref(_cdims(dims)), b"dec.1.xkv.w", C.byref(off), C.byref(numel)))
    assert np.array_equal(flat[off.value:off.value + numel.value], packed["dec.1.xkv.w"])
