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


def chunk_split(n_chunks, world):
    """chunks [c_g, c_{g+1}) owned by rank g -- must match shard_layout() in csrc/lcpc_hip.cpp."""
    return [(n_chunks * g // world, n_chunks * (g + 1) // world) for g in range(world)]


def aligned_nodes(c0, c1):
    """[c0, c1) as maximal aligned power-of-two blocks [(first_chunk, log2 size)] -- must match shard_nodes()
    in csrc/lcpc_hip.cpp (checked against lcpc_shard_nodes by tests/test_abi.py)."""
    out, pos = [], c0
    while pos < c1:
        l = 0
        while (pos == 0 or pos % (2 << l) == 0) and pos + (2 << l) <= c1:
            l += 1
        out.append((pos, l))
        pos += 1 << l
    return out


def slots_per_rank(n_chunks, world):
    return max(1, max(len(aligned_nodes(b, e)) for b, e in chunk_split(n_chunks, world)))


def exchange_nodes(local_nodes, n_chunks_total, group=None):
    """all-gather of per-rank [k_g, n_cols, 32] uint8 node CVs, padded to `slots` per rank; returns the raw
    gather buffer [world * slots, n_cols, 32] (rank g's nodes at rows g*slots ...) and `slots`."""
    world = dist.get_world_size(group)
    slots = slots_per_rank(n_chunks_total, world)
    n_cols = local_nodes.shape[1]
    if local_nodes.shape[0] == slots:
        pad = local_nodes.contiguous()
    else:
        pad = torch.zeros((slots, n_cols, 32), dtype=torch.uint8, device=local_nodes.device)
        pad[:local_nodes.shape[0]] = local_nodes
    if dist.get_backend(group) == "gloo" and pad.is_cuda:
        # debugging path (several ranks sharing one GPU, where RCCL refuses duplicate devices): stage through host
        host = torch.empty((world * slots, n_cols, 32), dtype=torch.uint8)
        dist.all_gather_into_tensor(host, pad.cpu(), group=group)
        flat = host.to(local_nodes.device)
    else:
        flat = torch.empty((world * slots, n_cols, 32), dtype=torch.uint8, device=local_nodes.device)
        dist.all_gather_into_tensor(flat, pad, group=group)  # rank g's block lands at rows [g*slots, (g+1)*slots)
    return flat, slots


class HipShardEngine:
    """product engine: lcpc_commit_shard_device / lcpc_commit_finish_device of include/lcpc_hip.h."""

    def __init__(self, enc):
        self.enc = enc
        self.rank, self.world = enc.params.shard_rank, max(1, enc.params.shard_count)

    def layout(self, n_rows_total):
        v = [C.c_uint64() for _ in range(5)]
        self.enc._check(_lib.lib().lcpc_shard_layout(self.enc._h, n_rows_total, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def commit_shard(self, local_coeffs, n_rows_total):
        rb, re, cb, ce, _ = self.layout(n_rows_total)
        n_nodes = len(aligned_nodes(cb, ce))
        nodes = torch.empty((n_nodes, self.enc.n_cols, 32), dtype=torch.uint8, device=local_coeffs.device)
        st = torch.cuda.current_stream().cuda_stream
        ptr = local_coeffs.data_ptr() if re > rb else None
        # a rank that owns no chunk still passes a valid (unused) output pointer
        out = nodes if nodes.numel() else torch.zeros(64, dtype=torch.uint8, device=local_coeffs.device)
        self.enc._check(_lib.lib().lcpc_commit_shard_device(self.enc._h, C.c_void_p(ptr), n_rows_total, C.c_void_p(st),
                                                            C.c_void_p(out.data_ptr())))
        return nodes

    def commit_finish(self, gathered, n_rows_total, slots, want_root=True):
        st = torch.cuda.current_stream().cuda_stream
        root = (C.c_uint8 * 32)() if want_root else None
        self.enc._check(_lib.lib().lcpc_commit_finish_device(self.enc._h, C.c_void_p(gathered.data_ptr()), n_rows_total, slots,
                                                             C.c_void_p(st), root))
        return bytes(root) if want_root else None


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


def sharded_prove(enc, outer_tensor, tr, group=None, allgather=None):
    """LcCommit::prove on the row-sharded commitment held by `enc`'s context (lcpc_prove_sharded).  Collective: every
    rank calls it with the same outer_tensor (n_rows_total x L) and an identical transcript; returns (proof bytes,
    opened columns), identical on every rank.  `allgather(send, recv, nbytes)` defaults to torch.distributed."""
    import numpy as np
    lib = _lib.lib()
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
    rc = lib.lcpc_prove_sharded(enc._h, t.ctypes.data_as(C.c_void_p), t.shape[0], tr._h, C.c_void_p(send.data_ptr()),
                                C.c_void_p(recv.data_ptr()), send.numel(), fn, None, C.byref(pp), C.byref(plen),
                                cols.ctypes.data_as(C.c_void_p))
    if err:
        raise err[0]
    enc._check(rc)
    data = C.string_at(pp, plen.value)
    lib.lcpc_free(pp)
    return data, cols


def sharded_commit(engine, local_coeffs, n_rows_total, group=None, want_root=True):
    """one row-sharded commit step: local encode + node CVs, all-gather, finish.  Returns the root."""
    _, _, _, _, n_chunks = engine.layout(n_rows_total)
    nodes = engine.commit_shard(local_coeffs, n_rows_total)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        gathered, slots = exchange_nodes(nodes, n_chunks, group)
    else:
        gathered, slots = nodes.contiguous(), max(1, nodes.shape[0])
    return engine.commit_finish(gathered, n_rows_total, slots, want_root)
