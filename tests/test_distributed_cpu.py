"""N > 1 path on CPU: world_size-2 (and 3) gloo process groups run lcpc_amd.distributed.sharded_commit with a
stand-in engine built on the *oracle* (the product engine is HIP-only), so the shard layout, the padded
all-gather of chunk chaining values and the reassembly order are exercised end to end and must reproduce the
unsharded oracle root.  Also checks chunk_split() against the C library's lcpc_shard_layout arithmetic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
import struct

import pyref as P
from lcpc_amd.distributed import aligned_nodes, chunk_split, sharded_commit, slots_per_rank


class OracleShardEngine:
    """CPU stand-in for HipShardEngine: same three methods, oracle arithmetic."""

    def __init__(self, fid, n_per_row, n_cols, rank, world):
        self.fid, self.L = fid, O.limbs(fid)
        self.enc = O.Encoding.ligero_from_dims(fid, n_per_row, n_cols)
        self.n_per_row, self.n_cols, self.rank, self.world = n_per_row, n_cols, rank, world
        self.elem_bytes = 8 * self.L
        self.hashes = None

    def layout(self, n_rows):
        F = 8 * self.L
        n_chunks = (32 + F * n_rows + 1023) // 1024
        cb, ce = chunk_split(n_chunks, self.world, F)[self.rank]

        def first_row(ch):
            return 0 if ch == 0 else min(n_rows, (ch * 1024 - 32) // F)
        rb = first_row(cb)
        re = n_rows if ce >= n_chunks else first_row(ce)
        if cb == ce:
            re = rb
        return rb, re, cb, ce, n_chunks

    def commit_shard(self, local_coeffs, n_rows):
        self.commit_encode(local_coeffs, n_rows)
        return self.commit_hash_cols(0, self.n_cols)

    def commit_finish(self, gathered, n_rows, slots, want_root=True):
        self.commit_finish_cols(gathered, slots, 0, self.n_cols)
        return self.commit_merkle(want_root)

    # ---- the steps behind the two calls ----
    def commit_encode(self, local_coeffs, n_rows, borrow=False):
        rb, re, cb, ce, _ = self.layout(n_rows)
        rows = local_coeffs.numpy().view(np.uint64).reshape(re - rb, self.n_per_row, self.L)
        comm = np.zeros((re - rb, self.n_cols, self.L), np.uint64)
        for r in range(re - rb):
            comm[r, :self.n_per_row] = rows[r]
            comm[r] = self.enc.encode(comm[r].copy()).reshape(self.n_cols, self.L)
        self._n_rows = n_rows
        self._cvs = O.leaf_chunk_cvs(self.fid, comm, self.n_cols, rb, re - rb, n_rows, cb, ce)
        np2 = 1 << max(0, (self.n_cols - 1).bit_length())
        self.hashes = np.zeros((2 * np2 - 1, 32), np.uint8)

    def commit_hash_cols(self, c0, c1):
        _, _, cb, ce, _ = self.layout(self._n_rows)
        cvs = self._cvs
        # pre-merge the chunk CVs into aligned subtree nodes (BLAKE3 parent rule, no ROOT), as the HIP engine does
        nodes = []
        _, _, _, _, n_chunks = self.layout(self._n_rows)
        an = aligned_nodes(cb, ce)
        whole = len(an) == 1 and cb == 0 and ce == n_chunks      # this rank's one node is the whole message: its last parent is the root
        for first, lg in an:
            out = np.zeros((c1 - c0, 32), np.uint8)
            for col in range(c0, c1):
                cur = [struct.unpack("<8I", cvs[first - cb + i, col].tobytes()) for i in range(1 << lg)]
                while len(cur) > 1:
                    cur = [P.b3_parent(cur[2 * i], cur[2 * i + 1], whole and len(cur) == 2) for i in range(len(cur) // 2)]
                out[col - c0] = np.frombuffer(struct.pack("<8I", *cur[0]), np.uint8)
            nodes.append(out)
        arr = np.stack(nodes) if nodes else np.zeros((0, c1 - c0, 32), np.uint8)
        return torch.from_numpy(arr)

    def commit_finish_cols(self, gathered, slots, c0, c1):
        """fold the gathered nodes of every rank (in chunk order) with the BLAKE3 stack rule into the leaf digests of [c0, c1)"""
        _, _, _, _, n_chunks = self.layout(self._n_rows)
        g = gathered.numpy()
        order = []
        for r, (b, e) in enumerate(chunk_split(n_chunks, self.world, self.elem_bytes)):
            for k, (first, lg) in enumerate(aligned_nodes(b, e)):
                order.append((r * slots + k, lg))
        for col in range(c0, c1):
            if n_chunks == 1 or len(order) == 1:
                self.hashes[col] = g[order[0][0], col - c0]
                continue
            stack, total = [], 0
            for j, (slot, lg) in enumerate(order):
                cv = struct.unpack("<8I", g[slot, col - c0].tobytes())
                if j == len(order) - 1:
                    break
                total += 1 << lg
                t = total >> lg
                while t & 1 == 0:
                    cv = P.b3_parent(stack.pop(), cv, False)
                    t >>= 1
                stack.append(cv)
            while stack:
                left = stack.pop()
                cv = P.b3_parent(left, cv, len(stack) == 0)
            self.hashes[col] = np.frombuffer(struct.pack("<8I", *cv), np.uint8)

    def commit_merkle(self, want_root=True):
        hashes = self.hashes
        np2 = (len(hashes) + 1) // 2
        width, ins, outs = np2, 0, np2
        while width > 1:
            for i in range(width // 2):
                hashes[outs + i] = np.frombuffer(O.blake3(hashes[ins + 2 * i].tobytes() + hashes[ins + 2 * i + 1].tobytes()), np.uint8)
            ins, outs, width = outs, outs + width // 2, width // 2
        return hashes[-1].tobytes()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fid, n_rows, n_per_row, n_cols, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = OracleShardEngine(fid, n_per_row, n_cols, rank, world)
        rb, re, _, _, _ = eng.layout(n_rows)
        coeffs = O.random_elems(fid, n_rows * n_per_row, 17).reshape(n_rows, n_per_row, -1)
        local = torch.from_numpy(coeffs[rb:re].copy().view(np.int64))
        root = sharded_commit(eng, local, n_rows)
        q.put((rank, root, eng.hashes.tobytes()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,fid,n_rows,n_per_row,n_cols", [
    (2, 3, 70, 32, 64),      # ft255: 3 chunks over 2 ranks (uneven), rows 0..62 | 63..69
    (2, 0, 300, 16, 32),     # ft63: 128 rows per chunk
    (3, 3, 40, 16, 32),      # 2 chunks over 3 ranks: one rank owns nothing
    (2, 2, 300, 16, 32),     # ft191: 24-byte elements straddle chunks; 8 chunks, the cut moves from chunk 4 down to chunk 2 (row 84)
    (3, 2, 500, 16, 32),     # ft191: 12 chunks over 3 ranks: cuts at chunks 2 and 8
    (2, 2, 57, 16, 32),      # ft191: 2 chunks and no possible cut: rank 1 owns the whole message as ONE node, which must carry ROOT
])
def test_sharded_commit_gloo(world, fid, n_rows, n_per_row, n_cols):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fid, n_rows, n_per_row, n_cols, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # unsharded oracle commit of the same coefficients
    enc = O.Encoding.ligero_from_dims(fid, n_per_row, n_cols)
    coeffs = O.random_elems(fid, n_rows * n_per_row, 17)
    oc = O.Commit.commit(coeffs, enc)
    for rank, root, hashes in res:
        assert root == oc.get_root(), "rank %d" % rank
        assert hashes == oc.hashes().tobytes()


def test_aligned_nodes_cover_and_align():
    for c0 in range(0, 40):
        for c1 in range(c0, 70):
            nodes = aligned_nodes(c0, c1)
            pos = c0
            for first, lg in nodes:
                assert first == pos and first % (1 << lg) == 0
                pos += 1 << lg
            assert pos == c1
    # the layouts the 8-GPU bench uses at 2^26 (17 chunks): 1,1,1,1,1,1,1,2 nodes; 2 GPUs: 1 + 2
    assert [len(aligned_nodes(b, e)) for b, e in chunk_split(17, 8)] == [1, 1, 1, 1, 1, 1, 1, 2]
    assert [len(aligned_nodes(b, e)) for b, e in chunk_split(17, 2)] == [1, 2]
    assert slots_per_rank(17, 4) == 2 and slots_per_rank(33, 8) == 2


def test_chunk_split_is_a_partition():
    for n in range(1, 70):
        for w in (1, 2, 3, 4, 8):
            sp = chunk_split(n, w)
            assert sp[0][0] == 0 and sp[-1][1] == n
            assert all(sp[i][1] == sp[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in sp) - min(e - b for b, e in sp) <= 1


def _ag_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lcpc_amd.distributed import allgather_bytes
        send = torch.full((100,), rank + 1, dtype=torch.uint8)
        send[40:] = 99                                   # beyond nbytes: must not travel
        recv = torch.zeros(100 * world, dtype=torch.uint8)
        allgather_bytes(send, recv, 40)
        q.put((rank, recv.numpy().tobytes()))
    finally:
        dist.destroy_process_group()


def test_allgather_bytes_gloo():
    """the exchange primitive of sharded_prove (lcpc_prove_sharded's callback): rank g's first nbytes land at g*nbytes."""
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ag_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = bytes([1] * 40 + [2] * 40 + [3] * 40) + bytes(300 - 120)
    for _, got in res:
        assert got == want
