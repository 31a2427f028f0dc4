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
    """coeffs_rows: torch int64 cuda tensor [n_rows, n_per_row, L]"""
    engines = [HipShardEngine(mk_enc((g, G))) for g in range(G)]
    nodes = []
    for g, eng in enumerate(engines):
        rb, re, cb, ce, nch = eng.layout(n_rows)
        local = coeffs_rows[rb:re].contiguous()
        nodes.append(eng.commit_shard(local, n_rows))
        assert (cb, ce) == chunk_split(nch, G)[g]
        assert nodes[-1].shape[0] == len(aligned_nodes(cb, ce))
    # emulate the all-gather: equal-sized padded blocks, rank g at rows [g*slots, (g+1)*slots)
    slots = slots_per_rank(nch, G)
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
    (3, 20, 64, 128, 2),        # single chunk: the "CV" is already the digest
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
        c = LcCommit(eng.enc)
        assert (c.hashes() == ref.hashes()).all()
        rb, re, _, _, _ = eng.layout(n_rows)
        if re > rb:
            assert (c.comm(rb, re - rb) == oc.comm().reshape(n_rows, -1)[rb:re].reshape(-1, L)).all()


def test_sharding_rejects_straddling_field():
    # ft191 rows (24 B) straddle 1 KiB chunk boundaries: row sharding is refused rather than silently wrong
    with pytest.raises(lcpc_amd.LcpcError) as e:
        LigeroEncoding.new_from_dims(2, 64, 128, shard=(0, 2))
    assert e.value.code == lcpc_amd.ERR_ARG


@pytest.mark.parametrize("fid,n_per_row,n_rows,G", [(3, 300, 70, 2), (3, 300, 70, 4), (0, 400, 300, 2), (3, 257, 40, 8)])
def test_sharded_brakedown_equals_unsharded(oracle, fid, n_per_row, n_rows, G):
    """Brakedown shards the same way (rows independent, matrices replicated on every rank): shards with
    >= 16 local rows take the position-major SpMM path, smaller ones the row-major one."""
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
        c = LcCommit(eng.enc)
        assert (c.hashes() == oc.hashes()).all()
        rb, re, _, _, _ = eng.layout(n_rows)
        if re > rb:
            assert (c.comm(rb, re - rb) == oc.comm().reshape(n_rows, -1)[rb:re].reshape(-1, L)).all()
            assert (c.coeffs(rb, re - rb) == oc.coeffs().reshape(n_rows, -1)[rb:re].reshape(-1, L)).all()
