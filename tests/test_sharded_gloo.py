"""Multi-GPU path on CPU: utterance sharding and the single result gather, run
with world_size 2 over gloo (no collective on the search path,
one gather of the finished results)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kaldi_decoder import sharded


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 128, 1000, 1024):
        for world in (1, 2, 3, 8):
            spans = [sharded.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(0)
    paths = [(rng.integers(0, 500, n).astype(np.int32), rng.integers(0, 9, n).astype(np.int32))
             for n in (0, 1, 17, 300)]
    out = sharded.unpack_paths(sharded.pack_paths(paths))
    assert len(out) == len(paths)
    for (a, b), (c, d) in zip(paths, out):
        assert a.tolist() == c.tolist() and b.tolist() == d.tolist()
    assert sharded.unpack_paths(sharded.pack_paths([])) == []


def _fake_path(utt):
    rng = np.random.default_rng(utt)
    n = int(rng.integers(0, 40))
    return rng.integers(1, 500, n).astype(np.int32), rng.integers(0, 20000, n).astype(np.int32)


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharded.shard_range(total, rank, world)
    mine = [_fake_path(u) for u in range(lo, hi)]
    got = sharded.gather_paths(mine, dst=0)
    if rank == 0:
        ok = len(got) == total and all(
            got[u][0].tolist() == _fake_path(u)[0].tolist() and
            got[u][1].tolist() == _fake_path(u)[1].tolist() for u in range(total))
        q.put(ok)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 64])
def test_gather_paths_world2_gloo(total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_gather_without_process_group_is_identity():
    paths = [_fake_path(3), _fake_path(4)]
    assert sharded.gather_paths(paths) == paths


def test_blob_roundtrip_and_single_process_lattice_gather():
    rng = np.random.default_rng(1)
    blobs = [rng.integers(-5, 5, n).astype(np.int32) for n in (0, 3, 1000)]
    out = sharded.unpack_blobs(sharded.pack_blobs(blobs))
    assert [b.tolist() for b in out] == [b.tolist() for b in blobs]

    class Lat:  # the fields of cabi.RawLattice that lattice_blob reads
        num_states, start = 3, 0
        src = np.array([0, 1], np.int32); dst = np.array([1, 2], np.int32)
        ilabel = np.array([5, 6], np.int32); olabel = np.array([0, 9], np.int32)
        graph_cost = np.array([0.5, 1.5], np.float32); ac_cost = np.array([2.0, 3.0], np.float32)
        state_frame = np.array([0, 1, 2], np.int32); state_graph = np.array([7, 8, 9], np.int32)
        final_state = np.array([2], np.int32); final_graph = np.array([0.25], np.float32)

    blob = sharded.lattice_blob(Lat)
    assert blob[:4].tolist() == [3, 2, 1, 0] and blob.dtype == np.int32
    assert blob[4 + 8:4 + 10].view(np.float32).tolist() == [0.5, 1.5]
    got = sharded.gather_lattices([Lat, Lat])
    assert len(got) == 2 and got[0].tolist() == blob.tolist()
