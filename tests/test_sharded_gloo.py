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


def _fake_lattice(utt):
    """A RawLattice-shaped object with random contents (what lattice_blob reads)."""
    from kaldi_decoder.cabi import RawLattice
    rng = np.random.default_rng(100 + utt)
    ns, na = int(rng.integers(1, 60)), int(rng.integers(0, 200))
    nf = int(rng.integers(0, 4))
    i32 = lambda n, hi: rng.integers(0, hi, n).astype(np.int32)  # noqa: E731
    f32 = lambda n: rng.standard_normal(n).astype(np.float32)  # noqa: E731
    return RawLattice(True, ns, int(rng.integers(0, ns)), i32(na, ns), i32(na, ns), i32(na, 500),
                      i32(na, 20000), f32(na), f32(na), i32(ns, 250), i32(ns, 1 << 22),
                      i32(nf, ns), f32(nf))


_LAT_FIELDS = ("src", "dst", "ilabel", "olabel", "graph_cost", "ac_cost", "state_frame",
               "state_graph", "final_state", "final_graph")


def _same_lattice(a, b):
    return (a.ok == b.ok and a.num_states == b.num_states and a.start == b.start and
            all(getattr(a, k).dtype == getattr(b, k).dtype and
                getattr(a, k).tobytes() == getattr(b, k).tobytes() for k in _LAT_FIELDS))


def test_lattice_blob_roundtrip():
    for u in range(6):
        lat = _fake_lattice(u)
        assert _same_lattice(sharded.lattice_from_blob(sharded.lattice_blob(lat)), lat)


def _lattice_worker(rank, world, port, spans, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = spans[rank]
    got = sharded.gather_lattices([_fake_lattice(u) for u in range(lo, hi)], dst=0,
                                  as_lattices=True)
    if rank == 0:
        total = spans[-1][1]
        q.put(len(got) == total and all(_same_lattice(got[u], _fake_lattice(u)) for u in range(total)))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("spans", [((0, 7), (7, 9)),     # unequal shards
                                   ((0, 5), (5, 5)),     # rank 1 has nothing
                                   ((0, 0), (0, 6))])    # the destination has nothing
def test_gather_lattices_world2_gloo(spans):
    """The lattice gather of north_star ("RCCL only for gather of final lattices"): the blob
    branch of gather_paths, with shards of different sizes and an empty shard; rank 0 must
    reassemble every RawLattice field bit for bit, in rank order."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_lattice_worker, args=(r, 2, port, spans, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _fake_batch(lo, hi):
    """A cabi.RawLattices holding _fake_lattice(lo..hi-1), as LatticeBatch.raw_lattices() builds it."""
    from kaldi_decoder.cabi import RawLattices
    ls = [_fake_lattice(u) for u in range(lo, hi)]
    sizes = np.array([[l.num_states, len(l.src), len(l.final_state), l.start] for l in ls],
                     np.int64).reshape(len(ls), 4)
    so, ao, fo = (np.zeros(len(ls) + 1, np.int64) for _ in range(3))
    if len(ls):
        so[1:], ao[1:], fo[1:] = np.cumsum(sizes[:, 0]), np.cumsum(sizes[:, 1]), np.cumsum(sizes[:, 2])
    cat = lambda k, dt: (np.concatenate([getattr(l, k) for l in ls]) if ls else np.zeros(0, dt))  # noqa: E731
    arcs = tuple(cat(k, np.float32 if k.endswith("cost") else np.int32) for k in _LAT_FIELDS[:6])
    states = tuple(cat(k, np.int32) for k in _LAT_FIELDS[6:8])
    finals = (cat("final_state", np.int32), cat("final_graph", np.float32))
    return RawLattices(sizes, so, ao, fo, arcs, states, finals)


def test_lattice_batch_pack_roundtrip():
    for lo, hi in ((0, 9), (3, 4), (5, 5)):
        got = sharded.unpack_lattice_batch(sharded.pack_lattice_batch(_fake_batch(lo, hi)))
        assert len(got) == hi - lo
        for k, u in enumerate(range(lo, hi)):
            assert _same_lattice(got[k], _fake_lattice(u))


def _batch_worker(rank, world, port, spans, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = spans[rank]
    got = sharded.gather_lattices(_fake_batch(lo, hi), dst=0, as_lattices=True)
    if rank == 0:
        total = spans[-1][1]
        q.put(len(got) == total and all(_same_lattice(got[u], _fake_lattice(u)) for u in range(total)))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("spans", [((0, 6), (6, 11)), ((0, 4), (4, 4))])
def test_gather_of_whole_call_results_world2_gloo(spans):
    """What bench.py --gpus N sends: one packed blob per rank and call (the batch-wide arrays of
    LatticeBatch.raw_lattices()), reassembled on rank 0 into every utterance's RawLattice."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, spans, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _threaded_worker(rank, world, port, n_calls, q):
    """bench.py's multi-GPU path on CPU: every rank submits `n_calls` packed call results to a
    ResultGatherer (a worker thread runs the exact-size gathers in order) while the "decode"
    thread goes on; rank 0 gets every call's per-rank payloads bit for bit."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sent = []
    gat = sharded.ResultGatherer(dst=0, on_done=sent.append)

    def payload(r, c):  # ranks and calls of different sizes, one of them empty
        n_utts = 0 if (r == 2 and c == 1) else 1 + (3 * r + c) % 4
        lo = 1000 * r + 10 * c
        return sharded.pack_lattice_batch(_fake_batch(lo, lo + n_utts))

    for c in range(n_calls):
        gat.submit(torch.from_numpy(payload(rank, c)))
    got = gat.close()
    assert sent == list(range(n_calls))
    if rank == 0:
        ok = len(got) == n_calls
        for c in range(n_calls):
            ok = ok and len(got[c]) == world
            for r in range(world):
                ok = ok and got[c][r].numpy().tobytes() == payload(r, c).tobytes()
                lats = sharded.unpack_lattice_batch(got[c][r].numpy())
                lo = 1000 * r + 10 * c
                ok = ok and all(_same_lattice(lats[k], _fake_lattice(lo + k)) for k in range(len(lats)))
        q.put(ok)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_threaded_exact_size_gather_world4_gloo():
    """north_star: "RCCL over xGMI only for gather of final lattices" -- here over gloo, four
    ranks: exact sizes (no pad-to-max), grouped point-to-point to rank 0, run by a worker
    thread so that the thread that enqueues decode calls never waits for another rank."""
    world = 4
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_threaded_worker, args=(r, world, port, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_result_gatherer_single_process():
    gat = sharded.ResultGatherer()
    a, b = torch.arange(5, dtype=torch.int32), torch.zeros(0, dtype=torch.int32)
    gat.submit(a)
    gat.submit(b)
    got = gat.close()
    assert len(got) == 2 and got[0][0] is a and got[1][0] is b


def _fake_pool(lo, hi):
    """The output buffer and table a LatticeBatch with a lent buffer would hold after a call
    whose lattices are _fake_lattice(lo..hi-1) (regions in reverse order, as the kernel may
    place them in any order): include/kd_hip.h, kd_batch_raw_lattice_table."""
    ls = [_fake_lattice(u) for u in range(lo, hi)]
    words, table, pos = [], np.zeros((len(ls), 5), np.int64), 0
    for i in reversed(range(len(ls))):
        l = ls[i]
        st = np.stack([l.state_frame, l.state_graph], 1).astype(np.int32)
        fi = np.stack([l.final_state, l.final_graph.view(np.int32)], 1).astype(np.int32)
        ar = np.stack([l.src, l.dst, l.ilabel, l.olabel, l.graph_cost.view(np.int32),
                       l.ac_cost.view(np.int32)], 1).astype(np.int32)
        region = np.concatenate([st.reshape(-1), fi.reshape(-1), ar.reshape(-1)]).view(np.int64)
        table[i] = [pos, l.num_states, len(l.src), len(l.final_state), l.start]
        words.append(region)
        pos += len(region)
    return torch.from_numpy(np.concatenate(words + [np.zeros(7, np.int64)])), pos, table


def test_pool_payload_roundtrip():
    for lo, hi in ((0, 7), (3, 4)):
        buf, used, table = _fake_pool(lo, hi)
        got = sharded.unpack_pool(sharded.pool_payload(buf, used, table))
        assert len(got) == hi - lo
        for k, u in enumerate(range(lo, hi)):
            assert _same_lattice(got[k], _fake_lattice(u))
