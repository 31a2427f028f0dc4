"""Row-sharded commit on the GPU: G shard contexts on one device (one per would-be rank), the all-gather
emulated by concatenating their chunk-CV tensors in rank order -- must equal the unsharded HIP commit and the
oracle, for the layouts the 8-GPU bench uses.  (The real exchange is torch.distributed over RCCL; its
assembly logic is covered on CPU by tests/test_distributed_cpu.py.)"""
import numpy as np
import pytest
import torch

import lcpc_amd
from lcpc_amd import LcCommit, LigeroEncoding, SdigEncoding
from lcpc_amd.distributed import HipShardEngine, aligned_nodes, chunk_split, slots_per_rank

pytestmark = pytest.mark.gpu


def run_sharded(mk_enc, G, coeffs_rows, n_rows):
    """coeffs_rows: torch int64 cuda tensor [n_rows, n_per_row, L]: the two-call form (shard, finish) around an emulated all-gather"""
    engines = [HipShardEngine(mk_enc((g, G))) for g in range(G)]
    nodes = []
    for g, eng in enumerate(engines):
        rb, re, cb, ce, nch = eng.layout(n_rows)
        local = coeffs_rows[rb:re].contiguous()
        nodes.append(eng.commit_shard(local, n_rows))
        assert (cb, ce) == chunk_split(nch, G, eng.elem_bytes)[g]
        assert nodes[-1].shape[0] == len(aligned_nodes(cb, ce))
    # emulate the all-gather: equal-sized padded blocks, rank g at rows [g*slots, (g+1)*slots)
    slots = slots_per_rank(nch, G, engines[0].elem_bytes)
    gathered = torch.zeros((G * slots, nodes[0].shape[1], 32), dtype=torch.uint8, device="cuda")
    for g, nd in enumerate(nodes):
        gathered[g * slots:g * slots + nd.shape[0]] = nd
    roots = [eng.commit_finish(gathered.clone(), n_rows, slots) for eng in engines]
    return roots, engines


@pytest.mark.parametrize("fid,n_rows,n_per_row,n_cols,G", [
    (3, 512, 256, 512, 8),      # headline row count: 17 chunks over 8 ranks (2,2,2,2,2,2,2,3)
    (3, 512, 256, 512, 2),
    (3, 1024, 128, 256, 8),     # C4 row count: 33 chunks
    (3, 70, 64, 128, 4),        # 3 chunks over 4 ranks: one rank owns nothing
    (0, 300, 128, 256, 2),      # ft63: 128 rows per chunk
    (1, 200, 64, 128, 4),       # ft127
    (3, 100, 2048, 4096, 4),    # two-pass rows: K1s, canonical comm on every rank
    (0, 260, 4096, 8192, 3),    # ft63 on K1n (canonical comm)
    (1, 130, 2048, 4096, 2),    # ft127 on K1n
    (3, 20, 64, 128, 2),        # single chunk: the "CV" is already the digest
    # ft191: 24-byte elements straddle the 1 KiB chunks; shards begin where a chunk boundary is a row boundary (rows = 84 mod 128,
    # every third chunk): 700 rows = 17 chunks, cuts possible at chunks 2, 5, 8, 11, 14
    (2, 700, 64, 128, 2),
    (2, 700, 64, 128, 3),
    (2, 700, 64, 128, 8),       # more ranks than cuts: some ranks own nothing
    (2, 1500, 2048, 4096, 4),   # K1n rows (canonical comm), 36 chunks
    (2, 90, 64, 128, 2),        # 3 chunks: the only cut is chunk 2 = row 84
    (2, 40, 64, 128, 2),        # single chunk
    (2, 57, 64, 128, 4),        # 2 chunks, no cut possible: the last rank owns the whole message as one node (ROOT in the pre-merge)
    (2, 170, 64, 128, 1),       # "sharded" over one rank, 4 chunks = one node
    (3, 40, 64, 128, 1),        # the same for ft255 (2 chunks)
])
def test_sharded_equals_unsharded(oracle, fid, n_rows, n_per_row, n_cols, G):
    O = oracle
    L = O.limbs(fid)
    coeffs = O.random_elems(fid, n_rows * n_per_row, 19)
    dev = torch.from_numpy(coeffs.view(np.int64)).cuda().reshape(n_rows, n_per_row, L)
    roots, engines = run_sharded(lambda sh: LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, shard=sh), G, dev, n_rows)
    ref = LcCommit.commit(coeffs, LigeroEncoding.new_from_dims(fid, n_per_row, n_cols))
    oc = O.Commit.commit(coeffs, O.Encoding.ligero_from_dims(fid, n_per_row, n_cols), n_threads=4)
    assert ref.get_root() == oc.get_root()
    for r in roots:
        assert r == ref.get_root()
    # every rank ends with the full hashes array; its comm holds exactly its own rows
    for g, eng in enumerate(engines):
        c = eng.cm
        assert (c.hashes() == ref.hashes()).all()
        rb, re, _, _, _ = eng.layout(n_rows)
        if re > rb:
            assert (c.comm(rb, re - rb) == oc.comm().reshape(n_rows, -1)[rb:re].reshape(-1, L)).all()


@pytest.mark.parametrize("fid,n_rows,n_per_row,n_cols,G", [(3, 512, 256, 512, 8), (3, 512, 256, 512, 2), (3, 1024, 128, 256, 8), (3, 70, 64, 128, 4),
                                                     (0, 3000, 64, 128, 3), (3, 20, 64, 128, 2)])
def test_compact_exchange_layout(oracle, fid, n_rows, n_per_row, n_cols, G):
    """the layout the native RCCL exchange produces (lcpc_commit_sharded_device: one all-gather of node 0 of every rank,
    then one broadcast per extra node) assembled by hand: slot g = node 0 of rank g, extras in rank order from slot G on;
    lcpc_commit_finish_device with slots_per_rank = 0 must fold it to the oracle's tree on every rank."""
    O = oracle
    L = O.limbs(fid)
    coeffs = O.random_elems(fid, n_rows * n_per_row, 29)
    dev = torch.from_numpy(coeffs.view(np.int64)).cuda().reshape(n_rows, n_per_row, L)
    engines = [HipShardEngine(LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, shard=(g, G))) for g in range(G)]
    nodes = []
    for g, eng in enumerate(engines):
        rb, re, cb, ce, nch = eng.layout(n_rows)
        nodes.append(eng.commit_shard(dev[rb:re].contiguous(), n_rows).clone())
    extras = [nd[k] for nd in nodes for k in range(1, nd.shape[0])]
    gathered = torch.zeros((G + len(extras), n_cols, 32), dtype=torch.uint8, device="cuda")
    for g, nd in enumerate(nodes):
        if nd.shape[0]:
            gathered[g] = nd[0]
    for i, e in enumerate(extras):
        gathered[G + i] = e
    oc = O.Commit.commit(coeffs, O.Encoding.ligero_from_dims(fid, n_per_row, n_cols), n_threads=4)
    for eng in engines:
        assert eng.commit_finish(gathered.clone(), n_rows, 0) == oc.get_root()
        assert (eng.cm.hashes() == oc.hashes()).all()


def test_ft191_shard_boundaries_are_row_boundaries():
    """Ft191's 24-byte elements straddle the 1 KiB BLAKE3 chunks (round 3 refused to shard the field): a shard may begin only where
    32 + 24 r is a multiple of 1024, r = 84 (mod 128) -- chunk 2 (mod 3).  lcpc_shard_layout must hand out exactly such cuts, the
    ranks' row and chunk ranges must partition the commitment, and no rank's range may split a row."""
    for n_rows in (40, 90, 700, 1500, 5000):
        for G in (2, 3, 5, 8):
            seen_r = seen_c = 0
            for g in range(G):
                eng = HipShardEngine(LigeroEncoding.new_from_dims(2, 64, 128, shard=(g, G)))
                rb, re, cb, ce, nch = eng.layout(n_rows)
                assert nch == (32 + 24 * n_rows + 1023) // 1024
                assert (rb, cb) == (seen_r, seen_c) and re >= rb and ce >= cb
                if 0 < cb < nch:
                    assert cb % 3 == 2 and rb % 128 == 84 and 32 + 24 * rb == 1024 * cb
                assert (cb, ce) == chunk_split(nch, G, 24)[g]
                seen_r, seen_c = re, ce
            assert (seen_r, seen_c) == (n_rows, nch)


@pytest.mark.parametrize("fid,n_per_row,n_rows,G", [(3, 300, 70, 2), (3, 300, 70, 4), (0, 400, 300, 2), (3, 257, 40, 8), (2, 300, 300, 3)])
def test_sharded_brakedown_equals_unsharded(oracle, fid, n_per_row, n_rows, G):
    """Brakedown shards the same way (rows independent, matrices replicated on every rank): shards with
    >= 24 local rows take the position-major SpMM path, smaller ones the row-major one."""
    O = oracle
    L = O.limbs(fid)
    oenc = O.Encoding.sdig_from_dims(fid, n_per_row, 0, 11, 3)
    _, _, n_cols = oenc.get_dims(n_per_row)
    coeffs = O.random_elems(fid, n_rows * n_per_row, 23)
    dev = torch.from_numpy(coeffs.view(np.int64)).cuda().reshape(n_rows, n_per_row, L)
    roots, engines = run_sharded(lambda sh: SdigEncoding(fid, None, 11, 3, 0, sh, _dims=(n_per_row, n_cols)), G, dev, n_rows)
    oc = O.Commit.commit(coeffs, oenc, n_threads=4)
    for r in roots:
        assert r == oc.get_root()
    for eng in engines:
        c = eng.cm
        assert (c.hashes() == oc.hashes()).all()
        rb, re, _, _, _ = eng.layout(n_rows)
        if re > rb:
            assert (c.comm(rb, re - rb) == oc.comm().reshape(n_rows, -1)[rb:re].reshape(-1, L)).all()
            assert (c.coeffs(rb, re - rb) == oc.coeffs().reshape(n_rows, -1)[rb:re].reshape(-1, L)).all()


class ThreadAllGather:
    """all-gather between G threads of one process (one per shard context): stands in for RCCL in lcpc_prove_sharded."""

    def __init__(self, G):
        import threading
        self.G, self.bar, self.slots = G, threading.Barrier(G), [None] * G

    def make(self, g):
        def ag(send, recv, nbytes):
            torch.cuda.synchronize()
            self.slots[g] = send[:nbytes]
            self.bar.wait()
            for h in range(self.G):
                recv[h * nbytes:(h + 1) * nbytes].copy_(self.slots[h])
            torch.cuda.synchronize()
            self.bar.wait()
        return ag


@pytest.mark.parametrize("kind,fid,n_rows,n_per_row,n_cols,G", [
    ("ligero", 3, 512, 256, 512, 8),     # headline row count
    ("ligero", 3, 70, 64, 128, 4),       # one rank owns no rows
    ("ligero", 0, 300, 128, 256, 2),
    ("ligero", 1, 130, 2048, 4096, 2),   # ft127 on K1n: opened columns come out of a canonical comm on every rank
    ("ligero", 3, 20, 64, 128, 2),       # single chunk: rank 1 owns nothing
    ("sdig", 3, 70, 300, 0, 4),
    ("ligero", 2, 700, 64, 128, 3),      # ft191: shards cut at rows 84 (mod 128)
])
def test_sharded_prove_equals_unsharded(oracle, kind, fid, n_rows, n_per_row, n_cols, G):
    """lcpc_prove_sharded on G shard contexts (threads + an in-process all-gather): every rank returns the proof of the
    unsharded prover == the oracle's proof, byte for byte; the oracle verifier accepts it."""
    import threading
    from common import mk_transcript
    from lcpc_amd import Transcript
    from lcpc_amd.distributed import sharded_prove
    O = oracle
    L = O.limbs(fid)
    coeffs = O.random_elems(fid, n_rows * n_per_row, 23)
    dev = torch.from_numpy(coeffs.view(np.int64)).cuda().reshape(n_rows, n_per_row, L)
    if kind == "ligero":
        mk = lambda sh: LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, shard=sh)
        oenc = O.Encoding.ligero_from_dims(fid, n_per_row, n_cols)
    else:
        oenc = O.Encoding.sdig_from_dims(fid, n_per_row, 0, 11, 3)
        _, _, nc = oenc.get_dims(n_per_row)
        mk = lambda sh: SdigEncoding(fid, None, 11, 3, 0, sh, _dims=(n_per_row, nc))
    roots, engines = run_sharded(mk, G, dev, n_rows)
    oc = O.Commit.commit(coeffs, oenc, n_threads=4)
    root = oc.get_root()
    assert all(r == root for r in roots)
    outer = O.random_elems(fid, n_rows, 29)
    n_open = engines[0].enc.get_n_col_opens()
    opf, ocols = oc.prove(outer, oenc, mk_transcript(O.Transcript, root, n_open))
    tag = ThreadAllGather(G)
    out = [None] * G

    def work(g):
        try:
            out[g] = sharded_prove(engines[g], outer, mk_transcript(Transcript, root, n_open), allgather=tag.make(g))
        except Exception as e:                     # do not leave the other threads stuck at the barrier
            out[g] = e
            tag.bar.abort()

    th = [threading.Thread(target=work, args=(g,)) for g in range(G)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    for g in range(G):
        assert not isinstance(out[g], Exception), out[g]
        data, cols = out[g]
        assert data == opf, g
        assert (cols == np.asarray(ocols, np.uint64)).all()
    inner = O.random_elems(fid, n_per_row, 31)
    rc, _ = O.verify(oenc, root, outer, inner, out[0][0], mk_transcript(O.Transcript, root, n_open))
    assert rc == 0


def test_split_phase_state_errors():
    """the finish step of the split-phase form refuses to run before its shard step (LCPC_ERR_STATE) and with another row count than
    the shard step's (LCPC_ERR_ARG); after a finish the commit is done: a second finish is refused again"""
    import ctypes as C
    lib = lcpc_amd._lib.lib()
    eng = HipShardEngine(LigeroEncoding.new_from_dims(3, 64, 128, shard=(0, 2)))
    buf = torch.zeros((4, 128, 32), dtype=torch.uint8, device="cuda")
    p = C.c_void_p(buf.data_ptr())
    assert lib.lcpc_commit_finish_device(eng.cm._h, p, 40, 1, None, None) in (lcpc_amd.ERR_STATE, lcpc_amd.ERR_ARG)
    rb, re, _, _, _ = eng.layout(40)
    eng.commit_shard(torch.zeros(((re - rb) * 64, 4), dtype=torch.int64, device="cuda"), 40)
    assert lib.lcpc_commit_finish_device(eng.cm._h, p, 41, 1, None, None) == lcpc_amd.ERR_ARG
    assert lib.lcpc_commit_finish_device(eng.cm._h, p, 40, 2, None, None) == 0
    assert lib.lcpc_commit_finish_device(eng.cm._h, p, 40, 2, None, None) == lcpc_amd.ERR_STATE
