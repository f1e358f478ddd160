"""Utterance sharding across the GPUs of one node and the single result gather.

Utterances never interact and the graph is read-only, so the batch is split
contiguously over ranks, every rank holds its own copy of the graph, and there
is no collective on the search path.  The only exchange is the gather of the
finished results to rank 0: an all_gather of payload sizes followed by one
gather of the packed int32 payload (RCCL when the backend is "nccl", gloo in the
CPU tests).  SURVEY.md section 8(e).
"""

from __future__ import annotations

import numpy as np


def shard_range(total: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of `total` utterances owned by `rank`."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_paths(paths):
    """List of (ilabel, olabel) int32 arrays -> one int32 payload.

    Layout: [n_utts, len_0 .. len_{n-1}, ilabel_0.., olabel_0.., ilabel_1.., ...]."""
    lens = np.asarray([len(p[0]) for p in paths], np.int32)
    parts = [np.asarray([len(paths)], np.int32), lens]
    for il, ol in paths:
        parts.append(np.asarray(il, np.int32))
        parts.append(np.asarray(ol, np.int32))
    return np.concatenate(parts) if parts else np.zeros(1, np.int32)


def pack_blobs(blobs):
    """List of int32 arrays -> [n, len_0 .. len_{n-1}, blob_0, blob_1, ...]."""
    lens = np.asarray([len(b) for b in blobs], np.int32)
    return np.concatenate([np.asarray([len(blobs)], np.int32), lens] +
                          [np.asarray(b, np.int32) for b in blobs])


def unpack_blobs(buf):
    n = int(buf[0])
    lens = buf[1:1 + n]
    pos = 1 + n
    out = []
    for k in lens:
        out.append(buf[pos:pos + int(k)].copy())
        pos += int(k)
    return out


def unpack_paths(buf):
    n = int(buf[0])
    lens = buf[1:1 + n]
    pos = 1 + n
    out = []
    for k in lens:
        k = int(k)
        out.append((buf[pos:pos + k].copy(), buf[pos + k:pos + 2 * k].copy()))
        pos += 2 * k
    return out


def lattice_blob(lat):
    """One int32 array holding a raw lattice (cabi.RawLattice): sizes, arcs, finals."""
    head = np.asarray([lat.num_states, len(lat.src), len(lat.final_state), lat.start], np.int32)
    return np.concatenate([head, lat.src, lat.dst, lat.ilabel, lat.olabel,
                           lat.graph_cost.view(np.int32), lat.ac_cost.view(np.int32),
                           lat.state_frame, lat.state_graph, lat.final_state,
                           lat.final_graph.view(np.int32)]).astype(np.int32, copy=False)


def lattice_from_blob(blob):
    """Inverse of lattice_blob: a cabi.RawLattice whose arrays are views into `blob`."""
    from .cabi import RawLattice
    blob = np.asarray(blob, np.int32)
    ns, na, nf, start = (int(x) for x in blob[:4])
    pos = 4
    fields = []
    for n in (na, na, na, na, na, na, ns, ns, nf, nf):
        fields.append(blob[pos:pos + n])
        pos += n
    src, dst, il, ol, gc, ac, sf, sg, fs, fg = fields
    return RawLattice(ns > 0, ns, start, src, dst, il, ol, gc.view(np.float32), ac.view(np.float32),
                      sf, sg, fs, fg.view(np.float32))


def pack_lattice_batch(lats):
    """cabi.RawLattices (the lattices of one decode call: batch-wide arrays plus offsets) ->
    one int32 blob, a dozen array operations whatever the number of utterances:
    [n, states, arcs, finals | sizes (4 per utterance) | the ten arrays back to back]."""
    n = len(lats)
    ns, na, nf = int(lats.state_off[-1]), int(lats.arc_off[-1]), int(lats.final_off[-1])
    head = np.asarray([n, ns, na, nf], np.int32)
    parts = [head, np.ascontiguousarray(lats.sizes, np.int32).reshape(-1)]
    parts += [np.ascontiguousarray(a[:na]).view(np.int32) for a in lats.arc_arrays]
    parts += [np.ascontiguousarray(a[:ns]).view(np.int32) for a in lats.state_arrays]
    parts += [np.ascontiguousarray(a[:nf]).view(np.int32) for a in lats.final_arrays]
    return np.concatenate(parts)


def unpack_lattice_batch(blob):
    """Inverse of pack_lattice_batch: a cabi.RawLattices of views into `blob`."""
    from .cabi import RawLattices
    blob = np.asarray(blob, np.int32)
    n, ns, na, nf = (int(x) for x in blob[:4])
    sizes = blob[4:4 + 4 * n].reshape(n, 4).astype(np.int64)
    pos = 4 + 4 * n
    arrs = []
    for k, m in enumerate((na,) * 6 + (ns,) * 2 + (nf,) * 2):
        a = blob[pos:pos + m]
        pos += m
        arrs.append(a.view(np.float32) if k in (4, 5, 9) else a)
    so, ao, fo = (np.zeros(n + 1, np.int64) for _ in range(3))
    so[1:], ao[1:], fo[1:] = np.cumsum(sizes[:, 0]), np.cumsum(sizes[:, 1]), np.cumsum(sizes[:, 2])
    return RawLattices(sizes, so, ao, fo, tuple(arrs[:6]), tuple(arrs[6:8]), tuple(arrs[8:]))


def gather_lattices(lats, dst=0, group=None, as_lattices=False, keep_on_device=False):
    """Gathers every rank's pruned raw lattices to `dst` in rank order -- the one exchange of
    a sharded decode.  `lats` is what LatticeBatch.raw_lattices() returned (packed as one
    blob per rank: pack_lattice_batch) or a list of RawLattice (one blob each).  On `dst`:
    a list of int32 blobs, or of RawLattice objects with `as_lattices`; with
    `keep_on_device` (RCCL) the gathered int32 tensors, one per rank, left in the
    destination GPU's memory (unpack_lattice_batch(t.cpu().numpy()) reads one).  None on
    the other ranks."""
    from .cabi import RawLattices
    if isinstance(lats, RawLattices):
        got = gather_paths([(pack_lattice_batch(lats), None)], dst=dst, group=group, _single=True,
                           _keep_on_device=keep_on_device)
        if got is None:
            return None
        if keep_on_device:  # (one int32 tensor per rank whatever the world size)
            import torch
            return [g if isinstance(g, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(g))
                    for g in got]
        if not as_lattices:
            return got
        return [lat for blob in got for lat in unpack_lattice_batch(blob)]
    if keep_on_device:
        raise ValueError("keep_on_device gathers the RawLattices of one decode call (one blob per rank)")
    got = gather_paths([(lattice_blob(lat), None) for lat in lats], dst=dst, group=group,
                       _single=True)
    if got is None or not as_lattices:
        return got
    return [lattice_from_blob(b) for b in got]


def pool_payload(words, used, table):
    """One call's exported lattices as ONE 1-D int64 tensor for the gather, built without
    touching the lattice data: [n utterances, words used | table (5 per utterance) | the used
    prefix of the output buffer].  `words` is the buffer the batch was lent
    (LatticeBatch.set_output_buffer: device memory under RCCL); the few header words are
    uploaded next to it."""
    import torch
    head = np.concatenate([np.asarray([len(table), used], np.int64), np.asarray(table, np.int64).reshape(-1)])
    return torch.cat([torch.from_numpy(head).to(words.device), words[:used]])


def unpack_pool(payload):
    """Inverse of pool_payload on the host: a cabi.RawLattices over the payload's words (an
    utterance's region: states {frame, graph state} | finals {state, weight} | arcs {src, dst |
    ilabel, olabel | graph, ac}, include/kd_hip.h)."""
    from .cabi import RawLattices
    buf = np.ascontiguousarray(payload.cpu().numpy() if hasattr(payload, "cpu") else payload, np.int64)
    n, used = int(buf[0]), int(buf[1])
    table = buf[2:2 + 5 * n].reshape(n, 5)
    pool = buf[2 + 5 * n:2 + 5 * n + used].view(np.int32).reshape(-1, 2)
    sizes = table[:, 1:5].copy()
    so, ao, fo = (np.zeros(n + 1, np.int64) for _ in range(3))
    so[1:], ao[1:], fo[1:] = np.cumsum(sizes[:, 0]), np.cumsum(sizes[:, 1]), np.cumsum(sizes[:, 2])
    ns, na, nf = int(so[-1]), int(ao[-1]), int(fo[-1])
    sf, sg, fs = (np.empty(m, np.int32) for m in (ns, ns, nf))
    fg = np.empty(nf, np.float32)
    src, dst, il, ol = (np.empty(na, np.int32) for _ in range(4))
    gc, ac = np.empty(na, np.float32), np.empty(na, np.float32)
    for i in range(n):
        off, s_n, a_n, f_n = int(table[i, 0]), int(sizes[i, 0]), int(sizes[i, 1]), int(sizes[i, 2])
        if s_n == 0 or off < 0:
            continue
        st = pool[off:off + s_n]
        fi = pool[off + s_n:off + s_n + f_n]
        ar = pool[off + s_n + f_n:off + s_n + f_n + 3 * a_n].reshape(a_n, 6)
        sf[so[i]:so[i + 1]], sg[so[i]:so[i + 1]] = st[:, 0], st[:, 1]
        fs[fo[i]:fo[i + 1]], fg[fo[i]:fo[i + 1]] = fi[:, 0], fi[:, 1].view(np.float32)
        a0, a1 = ao[i], ao[i + 1]
        src[a0:a1], dst[a0:a1], il[a0:a1], ol[a0:a1] = ar[:, 0], ar[:, 1], ar[:, 2], ar[:, 3]
        gc[a0:a1], ac[a0:a1] = ar[:, 4].view(np.float32), ar[:, 5].view(np.float32)
    return RawLattices(sizes, so, ao, fo, (src, dst, il, ol, gc, ac), (sf, sg), (fs, fg))


def gather_exact(payload, dst=0, group=None):
    """One 1-D tensor per rank to `dst`, exact sizes, no padding: an all_gather of the
    lengths, then point-to-point -- every other rank sends its tensor as it is, `dst` posts one
    receive of exactly that length per rank (grouped: batch_isend_irecv; on RCCL that is one
    grouped launch over the direct xGMI links, not a ring).  The tensor stays where it is: a
    device tensor goes out of device memory and arrives in `dst`'s device memory.  Returns the
    list of per-rank tensors on `dst` (its own payload included, not copied), None elsewhere."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [payload]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if payload.is_cuda and dist.get_backend(group) == "gloo":
        payload = payload.cpu()  # (gloo's wire is host memory; RCCL takes the device tensor as it is)
    n = torch.tensor([payload.numel()], dtype=torch.int64, device=payload.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    if rank != dst:
        if payload.numel():
            dist.send(payload, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        return None
    sizes = [int(x) for x in torch.cat(sizes).tolist()]  # (the one host read of the exchange)
    bufs = [payload if r == rank else torch.empty(sizes[r], dtype=payload.dtype, device=payload.device)
            for r in range(world)]
    ops = [dist.P2POp(dist.irecv, bufs[r], dist.get_global_rank(group, r) if group is not None else r, group)
           for r in range(world) if r != rank and sizes[r]]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return bufs


class ResultGatherer:
    """The gather of finished results, OFF the thread that enqueues decode calls.

    submit(payload) hands a call's packed results (a 1-D tensor: device memory under RCCL)
    to a worker thread and returns at once; the worker runs gather_exact for the calls in the
    order they were submitted -- every rank submits the same number of calls in the same
    order, so the collectives pair up -- on a stream of its own when the payload is on a GPU,
    so neither the host thread nor the search stream waits for another rank.  `on_done(seq)`
    is called (on the worker) when call `seq`'s payload has been sent: the buffer may be
    reused.  close() waits for everything and returns rank `dst`'s list of per-call results
    (each a list of per-rank tensors), None on the other ranks."""

    def __init__(self, dst=0, group=None, keep=True, on_done=None):
        import queue
        import threading
        self.dst, self.group, self.keep, self.on_done = dst, group, keep, on_done
        self.q = queue.Queue()
        self.results = []
        self.error = None
        self.n = 0
        self.seconds = []  # (per call: the exchange as the worker saw it, waits included)
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        import time

        import torch
        stream = None
        try:
            while True:
                item = self.q.get()
                if item is None:
                    return
                seq, payload, ready = item
                t0 = time.perf_counter()
                if payload.is_cuda:
                    if stream is None:
                        torch.cuda.set_device(payload.device)
                        stream = torch.cuda.Stream(device=payload.device)
                    with torch.cuda.stream(stream):
                        if ready is not None:
                            stream.wait_event(ready)  # (the export is on the search stream)
                        got = gather_exact(payload, self.dst, self.group)
                    stream.synchronize()
                else:
                    got = gather_exact(payload, self.dst, self.group)
                self.seconds.append(time.perf_counter() - t0)
                if self.keep and got is not None:
                    self.results.append(got)
                if self.on_done is not None:
                    self.on_done(seq)
                self.q.task_done()
        except BaseException as e:  # (reported by close(): a dead worker must not hang the ranks silently)
            self.error = e

    def submit(self, payload, ready=None):
        seq = self.n
        self.n += 1
        self.q.put((seq, payload, ready))
        return seq

    def mean_ms(self):
        """Mean duration of an exchange on the worker thread, in ms (None before the first)."""
        return 1e3 * sum(self.seconds) / len(self.seconds) if self.seconds else None

    def flush(self):
        """Waits until everything submitted so far has been gathered."""
        while self.q.unfinished_tasks and self.t.is_alive() and self.error is None:
            import time
            time.sleep(0.0002)
        if self.error is not None:
            raise self.error

    def close(self):
        self.flush()
        self.q.put(None)
        self.t.join()
        if self.error is not None:
            raise self.error
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_rank(self.group) != self.dst:
            return None
        return self.results


def gather_paths(paths, dst=0, group=None, _single=False, _keep_on_device=False):
    """Gathers every rank's best paths to `dst` in rank order; returns the
    concatenated list there and None elsewhere."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [p[0] for p in paths] if _single else list(paths)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    payload = torch.from_numpy(pack_blobs([p[0] for p in paths]) if _single
                               else pack_paths(paths)).to(dev)
    size = torch.tensor([payload.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    padded = torch.zeros(mx, dtype=torch.int32, device=dev)
    padded[:payload.numel()] = payload
    if rank == dst:
        bufs = [torch.zeros(mx, dtype=torch.int32, device=dev) for _ in range(world)]
        dist.gather(padded, bufs, dst=dst, group=group)
        if _keep_on_device:  # (each rank's payload: [1, length, blob]: strip the framing)
            return [bufs[r][2:sizes[r]] for r in range(world)]
        out = []
        for r in range(world):
            buf = bufs[r][:sizes[r]].cpu().numpy()
            out.extend(unpack_blobs(buf) if _single else unpack_paths(buf))
        return out
    dist.gather(padded, None, dst=dst, group=group)
    return None
