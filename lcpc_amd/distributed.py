"""Row-sharded commit across the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The commitment matrix is split by rows into blocks aligned to the 1 KiB BLAKE3 chunk boundaries of the leaf
message `0^32 || col[0] || col[1] || ...` (lcpc-2d/src/lib.rs:719-735), so each rank can encode its rows and
reduce them to chunk chaining values with no communication.  Before anything crosses the wire a rank merges
its chunk range into *aligned subtree nodes* (2^l consecutive chunks starting at a multiple of 2^l are a
subtree of the BLAKE3 tree), so it sends 1-2 CVs per column instead of one per chunk.  The single exchange
step of the path is ONE all-gather of those node CVs (n_cols x 32 B per node, equal-sized padded blocks);
every rank then folds the nodes into leaf digests straight out of the gather buffer (node table, no
re-packing copy) and builds the Merkle tree redundantly (~1 % of the work).  No other collective exists on the
commit path.  `sharded_prove` is the prover on such a commitment: collapse_columns splits by rows
(partial sums per rank, all-gather, sum mod p), open_column gathers each rank's rows of the opened columns
(one more all-gather); the transcript runs identically on every rank, so every rank ends with the same proof.

The compute is delegated to an `engine`, so that the exchange logic here can be exercised on CPU (gloo,
world_size 2/3) with a stand-in engine from the tests while the product engine is HIP:
    layout(n_rows_total) -> (row_begin, row_end, chunk_begin, chunk_end, n_chunks_total)
    commit_shard(local_coeffs, n_rows_total) -> uint8 tensor [n_nodes_of_this_rank, n_cols, 32]
    commit_finish(gathered [world * slots, n_cols, 32], n_rows_total, slots) -> 32-byte root
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib


def chunk_split(n_chunks, world, elem_bytes=32):
    """chunks [c_g, c_{g+1}) owned by rank g -- must match shard_chunk_range() in csrc/shard.cpp.  A shard begins where a chunk
    boundary of the leaf message is also a row boundary, (1024 c - 32) % elem_bytes == 0: every chunk when the element size divides
    1024 (the even split n g / world), every third chunk for Ft191's 24 bytes (the even split moved down to the nearest one)."""
    def bound(i):
        if i == 0:
            return 0
        if i >= world:
            return n_chunks
        c = n_chunks * i // world
        while c > 0 and (c * 1024 - 32) % elem_bytes:
            c -= 1
        return c
    if world <= 1:
        return [(0, n_chunks)]
    return [(bound(g), bound(g + 1)) for g in range(world)]


def aligned_nodes(c0, c1):
    """[c0, c1) as maximal aligned power-of-two blocks [(first_chunk, log2 size)] -- must match shard_nodes()
    in csrc/shard.cpp (checked against lcpc_shard_nodes by tests/test_abi.py)."""
    out, pos = [], c0
    while pos < c1:
        l = 0
        while (pos == 0 or pos % (2 << l) == 0) and pos + (2 << l) <= c1:
            l += 1
        out.append((pos, l))
        pos += 1 << l
    return out


def slots_per_rank(n_chunks, world, elem_bytes=32):
    return max(1, max(len(aligned_nodes(b, e)) for b, e in chunk_split(n_chunks, world, elem_bytes)))


_xchg_cache = {}


def exchange_nodes(local_nodes, n_chunks_total, group=None, elem_bytes=32):
    """all-gather of per-rank [k_g, n_cols, 32] uint8 node CVs, padded to `slots` per rank; returns the raw
    gather buffer [world * slots, n_cols, 32] (rank g's nodes at rows g*slots ...) and `slots`.  The pad / gather
    buffers are allocated once per shape and reused (the finish phase clobbers the gather buffer, nothing keeps it)."""
    world = dist.get_world_size(group)
    slots = slots_per_rank(n_chunks_total, world, elem_bytes)
    n_cols = local_nodes.shape[1]
    key = (world, slots, n_cols, str(local_nodes.device), dist.get_backend(group))
    bufs = _xchg_cache.get(key)
    if bufs is None:
        _xchg_cache.clear()
        host = local_nodes.is_cuda and dist.get_backend(group) == "gloo"
        bufs = {"pad": torch.zeros((slots, n_cols, 32), dtype=torch.uint8, device=local_nodes.device),
                "flat": torch.empty((world * slots, n_cols, 32), dtype=torch.uint8, device=local_nodes.device),
                "host": torch.empty((world * slots, n_cols, 32), dtype=torch.uint8) if host else None}
        _xchg_cache[key] = bufs
    if local_nodes.shape[0] == slots and local_nodes.is_contiguous():
        pad = local_nodes
    else:
        pad = bufs["pad"]
        pad[:local_nodes.shape[0]] = local_nodes
    flat = bufs["flat"]
    if bufs["host"] is not None:
        # debugging path (several ranks sharing one GPU, where RCCL refuses duplicate devices): stage through host
        dist.all_gather_into_tensor(bufs["host"], pad.cpu(), group=group)
        flat.copy_(bufs["host"])
    else:
        dist.all_gather_into_tensor(flat, pad, group=group)  # rank g's block lands at rows [g*slots, (g+1)*slots)
    return flat, slots


class HipShardEngine:
    """product engine: one sharded encoder context + one LcCommit object (include/lcpc_hip.h).  Two exchange modes:
    `commit_shard` / `commit_finish` around a caller-side all-gather (torch.distributed; gloo in the CPU tests), or, after
    `comm_init`, `commit_native`: shard -> ncclAllGather -> finish inside the library on one stream."""

    def __init__(self, enc):
        from . import LcCommit
        self.enc = enc
        self.cm = LcCommit(enc)
        self.rank, self.world = enc.params.shard_rank, max(1, enc.params.shard_count)
        self.n_cols = enc.n_cols
        self.elem_bytes = 8 * enc.L
        self._layout = {}
        self._nodes = None

    def layout(self, n_rows_total):
        if n_rows_total not in self._layout:
            v = [C.c_uint64() for _ in range(5)]
            self.enc._check(_lib.lib().lcpc_shard_layout(self.enc._h, n_rows_total, *[C.byref(x) for x in v]))
            self._layout[n_rows_total] = tuple(x.value for x in v)
        return self._layout[n_rows_total]

    def commit_shard(self, local_coeffs, n_rows_total, borrow=False):
        rb, re, cb, ce, _ = self.layout(n_rows_total)
        n_nodes = len(aligned_nodes(cb, ce))
        if self._nodes is None or self._nodes.shape[0] != max(n_nodes, 1) or self._nodes.device != local_coeffs.device:
            # (a rank that owns no chunk still passes a valid, unused, output pointer)
            self._nodes = torch.empty((max(n_nodes, 1), self.enc.n_cols, 32), dtype=torch.uint8, device=local_coeffs.device)
        st = torch.cuda.current_stream().cuda_stream
        ptr = local_coeffs.data_ptr() if re > rb else None
        self.cm._check(_lib.lib().lcpc_commit_shard_device(self.cm._h, C.c_void_p(ptr), n_rows_total, C.c_void_p(st),
                                                           1 if borrow else 0, C.c_void_p(self._nodes.data_ptr())))
        return self._nodes[:n_nodes]

    def commit_finish(self, gathered, n_rows_total, slots, want_root=True):
        st = torch.cuda.current_stream().cuda_stream
        root = (C.c_uint8 * 32)() if want_root else None
        self.cm._check(_lib.lib().lcpc_commit_finish_device(self.cm._h, C.c_void_p(gathered.data_ptr()), n_rows_total, slots,
                                                            C.c_void_p(st), root))
        if want_root:
            self.cm._refresh()
        return bytes(root) if want_root else None

    # ---- native RCCL exchange (lcpc_comm_init / lcpc_commit_sharded_device / lcpc_prove_sharded_rccl) ----
    def comm_init(self, group=None):
        """bring up the library's own RCCL communicator: rank 0 draws an ncclUniqueId, torch.distributed carries the 128
        bytes to the other ranks (control plane only), every rank calls lcpc_comm_init."""
        lib = _lib.lib()
        idb = (C.c_uint8 * 128)()
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        # every rank draws an id first: that proves its process can load and call RCCL.  ncclCommInitRank is a collective --
        # a rank that failed before reaching it would leave the others waiting -- so the outcome is agreed on before anyone
        # enters it (and rank 0's id is the one that is used)
        rc = lib.lcpc_comm_unique_id(idb)
        if world > 1:
            ok = torch.tensor([0 if rc else 1], dtype=torch.int32, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0:
                from . import LcpcError
                raise LcpcError(rc if rc else -20, "another rank could not load RCCL")
            box = [bytes(idb)]
            dist.broadcast_object_list(box, src=0, group=group)
            idb = (C.c_uint8 * 128).from_buffer_copy(box[0])
        elif rc:
            from . import LcpcError
            raise LcpcError(rc)
        self.enc._check(lib.lcpc_comm_init(self.enc._h, idb, rank, world))

    def commit_native(self, local_coeffs, n_rows_total, want_root=True, borrow=False, async_tail=False):
        """lcpc_commit_sharded_device.  async_tail: LCPC_COMMIT_ASYNC_TAIL -- the exchange, the leaf digests and the tree run on the
        commitment's own stream and the caller's stream is free again after the column hash (a second engine of the same encoder
        can then encode the next commitment while this one's node values are on the wire)."""
        rb, re, _, _, _ = self.layout(n_rows_total)
        st = torch.cuda.current_stream().cuda_stream
        root = (C.c_uint8 * 32)() if want_root else None
        ptr = local_coeffs.data_ptr() if re > rb else None
        self.cm._check(_lib.lib().lcpc_commit_sharded_device(self.cm._h, C.c_void_p(ptr), n_rows_total, C.c_void_p(st),
                                                             (1 if borrow else 0) | (2 if async_tail else 0), root))
        if want_root:
            self.cm._refresh()
        return bytes(root) if want_root else None

    def exchange_probe(self):
        """lcpc_shard_exchange_probe: the exchange of the last commit_native alone, once more, enqueued on the current stream;
        returns the bytes this rank receives per exchange"""
        st = torch.cuda.current_stream().cuda_stream
        b = C.c_uint64()
        self.cm._check(_lib.lib().lcpc_shard_exchange_probe(self.cm._h, C.c_void_p(st), C.byref(b)))
        return b.value

    @staticmethod
    def rccl_version():
        """ncclGetVersion of the communicator library the native exchange loaded (0: none exported), None without a library"""
        v = C.c_int()
        return v.value if _lib.lib().lcpc_comm_rccl_version(C.byref(v)) == 0 else None

    def prove_native(self, outer_tensor, tr):
        import numpy as np
        t = np.ascontiguousarray(outer_tensor, np.uint64).reshape(-1, self.enc.L)
        pp, plen = C.c_void_p(), C.c_uint64()
        cols = np.zeros(self.enc.get_n_col_opens(), np.uint64)
        self.cm._check(_lib.lib().lcpc_prove_sharded_rccl(self.cm._h, t.ctypes.data_as(C.c_void_p), t.shape[0], tr._h, C.byref(pp),
                                                          C.byref(plen), cols.ctypes.data_as(C.c_void_p)))
        data = C.string_at(pp, plen.value)
        _lib.lib().lcpc_free(pp)
        return data, cols


def allgather_bytes(send, recv, nbytes, group=None):
    """all-gather the first nbytes of the uint8 device tensor `send` into `recv` (rank g at [g*nbytes, (g+1)*nbytes))."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        recv[:nbytes].copy_(send[:nbytes])
    elif dist.get_backend(group) == "gloo" and send.is_cuda:       # debugging path, see exchange_nodes
        host = torch.empty(world * nbytes, dtype=torch.uint8)
        dist.all_gather_into_tensor(host, send[:nbytes].cpu(), group=group)
        recv[:world * nbytes].copy_(host)
    else:
        dist.all_gather_into_tensor(recv[:world * nbytes], send[:nbytes], group=group)
    if send.is_cuda:
        torch.cuda.synchronize()


def sharded_prove(engine, outer_tensor, tr, group=None, allgather=None):
    """LcCommit::prove on the row-sharded commitment held by `engine` (lcpc_prove_sharded).  Collective: every
    rank calls it with the same outer_tensor (n_rows_total x L) and an identical transcript; returns (proof bytes,
    opened columns), identical on every rank.  `allgather(send, recv, nbytes)` defaults to torch.distributed."""
    import numpy as np
    lib = _lib.lib()
    enc, cm = engine.enc, engine.cm
    t = np.ascontiguousarray(outer_tensor, np.uint64).reshape(-1, enc.L)
    world = max(1, enc.params.shard_count)
    nb = int(lib.lcpc_prove_sharded_bytes(enc._h, t.shape[0]))
    dev = torch.device("cuda", enc.params.device)
    send = torch.empty(max(nb, 64), dtype=torch.uint8, device=dev)
    recv = torch.empty(max(nb, 64) * world, dtype=torch.uint8, device=dev)
    ag = allgather or (lambda s, r, n: allgather_bytes(s, r, n, group))
    err = []

    def cb(_user, nbytes):
        try:
            ag(send, recv, int(nbytes))
            return 0
        except Exception as e:      # never unwind through the C frame
            err.append(e)
            return 1

    fn = _lib.ALLGATHER_FN(cb)
    pp, plen = C.c_void_p(), C.c_uint64()
    cols = np.zeros(enc.get_n_col_opens(), np.uint64)
    rc = lib.lcpc_prove_sharded(cm._h, t.ctypes.data_as(C.c_void_p), t.shape[0], tr._h, C.c_void_p(send.data_ptr()),
                                C.c_void_p(recv.data_ptr()), send.numel(), fn, None, C.byref(pp), C.byref(plen),
                                cols.ctypes.data_as(C.c_void_p))
    if err:
        raise err[0]
    cm._check(rc)
    data = C.string_at(pp, plen.value)
    lib.lcpc_free(pp)
    return data, cols


def sharded_commit(engine, local_coeffs, n_rows_total, group=None, want_root=True, borrow=False):
    """one row-sharded commit step: local encode + node CVs, all-gather, finish.  Returns the root."""
    _, _, _, _, n_chunks = engine.layout(n_rows_total)
    eb = getattr(engine, "elem_bytes", 32)
    nodes = engine.commit_shard(local_coeffs, n_rows_total, borrow) if borrow else engine.commit_shard(local_coeffs, n_rows_total)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        gathered, slots = exchange_nodes(nodes, n_chunks, group, eb)
    else:
        gathered, slots = nodes.contiguous(), max(1, nodes.shape[0])
    return engine.commit_finish(gathered, n_rows_total, slots, want_root)
