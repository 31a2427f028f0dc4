"""Row-sharded commit across the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The commitment matrix is split by rows into blocks aligned to the 1 KiB BLAKE3 chunk boundaries of the leaf
message `0^32 || col[0] || col[1] || ...` (lcpc-2d/src/lib.rs:719-735), so each rank can encode its rows and
reduce them to ONE 32-byte chaining value per (chunk, column) with no communication.  The single exchange step
of the path is an all-gather of those chaining values (n_cols x 32 B per chunk); every rank then folds the
chunk CVs into leaf digests and builds the Merkle tree redundantly (it is ~1% of the work).  No other
collective exists on the commit path.

The compute is delegated to an `engine` with three methods, so that the exchange/assembly logic here can be
exercised on CPU (gloo, world_size 2) with a stand-in engine from the tests while the product engine is HIP:
    layout(n_rows_total) -> (row_begin, row_end, chunk_begin, chunk_end, n_chunks_total)
    commit_shard(local_coeffs, n_rows_total) -> torch.uint8 tensor [(chunk_end-chunk_begin), n_cols, 32]
    commit_finish(all_cvs [n_chunks_total, n_cols, 32], n_rows_total) -> 32-byte root
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib


def chunk_split(n_chunks, world):
    """chunks [c_g, c_{g+1}) owned by rank g -- must match shard_layout() in csrc/lcpc_hip.cpp."""
    return [(n_chunks * g // world, n_chunks * (g + 1) // world) for g in range(world)]


def exchange_chunk_cvs(local_cvs, n_chunks_total, group=None):
    """all-gather of per-rank [k_g, n_cols, 32] uint8 tensors (k_g differs by at most one between ranks)
    into the full [n_chunks_total, n_cols, 32] tensor, identical on every rank."""
    world = dist.get_world_size(group)
    split = chunk_split(n_chunks_total, world)
    kmax = max(e - b for b, e in split)
    n_cols = local_cvs.shape[1]
    pad = torch.zeros((kmax, n_cols, 32), dtype=torch.uint8, device=local_cvs.device)
    pad[:local_cvs.shape[0]] = local_cvs
    flat = torch.empty((world * kmax, n_cols, 32), dtype=torch.uint8, device=local_cvs.device)
    dist.all_gather_into_tensor(flat, pad, group=group)      # rank g's block lands at rows [g*kmax, (g+1)*kmax)
    if all(e - b == kmax for b, e in split):
        return flat
    out = flat.view(world, kmax, n_cols, 32)
    return torch.cat([out[g, :e - b] for g, (b, e) in enumerate(split)], dim=0)


class HipShardEngine:
    """product engine: lcpc_commit_shard_device / lcpc_commit_finish_device of include/lcpc_hip.h."""

    def __init__(self, enc):
        self.enc = enc

    def layout(self, n_rows_total):
        v = [C.c_uint64() for _ in range(5)]
        self.enc._check(_lib.lib().lcpc_shard_layout(self.enc._h, n_rows_total, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def commit_shard(self, local_coeffs, n_rows_total):
        rb, re, cb, ce, _ = self.layout(n_rows_total)
        cvs = torch.empty((max(ce - cb, 0), self.enc.n_cols, 32), dtype=torch.uint8, device=local_coeffs.device)
        st = torch.cuda.current_stream().cuda_stream
        ptr = local_coeffs.data_ptr() if local_coeffs.numel() else 0
        self.enc._check(_lib.lib().lcpc_commit_shard_device(self.enc._h, C.c_void_p(ptr), n_rows_total, C.c_void_p(st),
                                                            C.c_void_p(cvs.data_ptr() if cvs.numel() else 0) if cvs.numel() else C.c_void_p(self._dummy().data_ptr())))
        return cvs

    def _dummy(self):
        if not hasattr(self, "_d"):
            self._d = torch.zeros(64, dtype=torch.uint8, device="cuda")
        return self._d

    def commit_finish(self, all_cvs, n_rows_total, want_root=True):
        st = torch.cuda.current_stream().cuda_stream
        root = (C.c_uint8 * 32)() if want_root else None
        self.enc._check(_lib.lib().lcpc_commit_finish_device(self.enc._h, C.c_void_p(all_cvs.data_ptr()), n_rows_total,
                                                             C.c_void_p(st), root))
        return bytes(root) if want_root else None


def sharded_commit(engine, local_coeffs, n_rows_total, group=None, want_root=True):
    """one row-sharded commit step: local encode + chunk CVs, all-gather, finish.  Returns the root."""
    _, _, _, _, n_chunks = engine.layout(n_rows_total)
    cvs = engine.commit_shard(local_coeffs, n_rows_total)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        all_cvs = exchange_chunk_cvs(cvs, n_chunks, group)
    else:
        all_cvs = cvs
    return engine.commit_finish(all_cvs.contiguous(), n_rows_total, want_root)
