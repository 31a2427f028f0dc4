"""The library's native RCCL exchange (lcpc_comm_init / lcpc_commit_sharded_device / lcpc_prove_sharded_rccl).

world = 1 runs on the single-GPU box: communicator bring-up through ncclCommInitRank, shard -> ncclAllGather -> finish on
one (non-default) stream with no host synchronisation in between, and the three all-gathers of the sharded prove.  With
two or more GPUs visible the same flow runs as real processes over RCCL (tests/test_gpu_multiproc.py)."""
import numpy as np
import pytest
import torch

import lcpc_amd
from common import mk_transcript
from lcpc_amd import LcCommit, LigeroEncoding, SdigEncoding, Transcript
from lcpc_amd.distributed import HipShardEngine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,fid,n_rows,n_per_row,n_cols", [
    ("ligero", 3, 512, 256, 512),      # headline row count: 17 BLAKE3 chunks per leaf
    ("ligero", 3, 20, 64, 128),        # single chunk
    ("ligero", 0, 300, 128, 256),
    ("sdig", 3, 70, 300, 0),
    ("ligero", 3, 512, 2048, 4096),
    ("ligero", 3, 20, 4096, 8192),     # single chunk (the "node" is the digest)
    ("ligero", 0, 300, 2048, 4096),
    ("ligero", 2, 700, 128, 256),      # ft191 (24-byte elements straddle chunks; world 1: one shard)
    ("ligero", 3, 40, 64, 128),        # 2 chunks, world 1: the rank's one node IS the whole message and must carry ROOT
    ("ligero", 3, 100, 2048, 4096),    # 4 chunks, world 1
    ("sdig", 3, 70, 3000, 0),          # 4500-odd columns, position-major commitment
])
def test_native_exchange_world1(oracle, kind, fid, n_rows, n_per_row, n_cols):
    O = oracle
    L = O.limbs(fid)
    coeffs = O.random_elems(fid, n_rows * n_per_row, 61)
    if kind == "ligero":
        enc = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, shard=(0, 1))
        oenc = O.Encoding.ligero_from_dims(fid, n_per_row, n_cols)
    else:
        oenc = O.Encoding.sdig_from_dims(fid, n_per_row, 0, 11, 3)
        _, _, nc = oenc.get_dims(n_per_row)
        enc = SdigEncoding(fid, None, 11, 3, 0, (0, 1), _dims=(n_per_row, nc))
    eng = HipShardEngine(enc)
    with pytest.raises(lcpc_amd.LcpcError) as e:          # no communicator yet
        eng.commit_native(torch.zeros(8, dtype=torch.int64, device="cuda"), n_rows)
    assert e.value.code == lcpc_amd.ERR_STATE
    eng.comm_init()
    dev = torch.from_numpy(coeffs.view(np.int64)).cuda()
    oc = O.Commit.commit(coeffs, oenc, n_threads=4)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for borrow in (False, True):
            assert eng.commit_native(dev, n_rows, want_root=False, borrow=borrow) is None     # enqueue only
            root = eng.commit_native(dev, n_rows, want_root=True, borrow=borrow)
            assert root == oc.get_root()
    st.synchronize()
    assert (eng.cm.hashes() == oc.hashes()).all()
    assert (eng.cm.coeffs() == oc.coeffs()).all()
    outer = O.random_elems(fid, n_rows, 63)
    n_open = enc.get_n_col_opens()
    opf, ocols = oc.prove(outer, oenc, mk_transcript(O.Transcript, root, n_open))
    data, cols = eng.prove_native(outer, mk_transcript(Transcript, root, n_open))
    assert data == opf and (cols == np.asarray(ocols, np.uint64)).all()
    # the plain prover on the same (fully local) commitment gives the same bytes
    pf = eng.cm.prove(outer, enc, mk_transcript(Transcript, root, n_open))
    assert pf.to_bytes() == opf
    # re-initialising the communicator is allowed; destroying it disables the native path again
    eng.comm_init()
    assert eng.commit_native(dev, n_rows) == oc.get_root()
    enc._check(lcpc_amd._lib.lib().lcpc_comm_destroy(enc._h))
    with pytest.raises(lcpc_amd.LcpcError):
        eng.commit_native(dev, n_rows)


def test_comm_init_argument_checks():
    import ctypes as C
    lib = lcpc_amd._lib.lib()
    enc = LigeroEncoding.new_from_dims(3, 64, 128, shard=(1, 4))
    idb = (C.c_uint8 * 128)()
    assert lib.lcpc_comm_unique_id(idb) == 0
    assert lib.lcpc_comm_init(enc._h, idb, 0, 4) == lcpc_amd.ERR_ARG      # rank != shard_rank
    assert lib.lcpc_comm_init(enc._h, idb, 1, 2) == lcpc_amd.ERR_ARG      # world != shard_count
    assert lib.lcpc_comm_init(enc._h, idb, 5, 4) == lcpc_amd.ERR_ARG


def test_native_exchange_async_tail_two_commitments(oracle):
    """LCPC_COMMIT_ASYNC_TAIL: exchange, leaf digests and tree on the commitment's own stream, the caller's stream free after the
    column hash.  Two commitments of ONE sharded encoder are filled alternately, back to back, with different polynomials and no
    host synchronisation in between (the second one's encode overlaps the first one's exchange; their collectives share the
    communicator and must keep their order); then a refill of each (which has to wait for the object's own tail).  Roots, whole
    `hashes`, coeffs and proof bytes of both equal the oracle's for the LAST polynomial committed into each."""
    O, fid, n_rows, n_per_row, n_cols = oracle, 3, 512, 2048, 4096
    enc = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, shard=(0, 1))
    oenc = O.Encoding.ligero_from_dims(fid, n_per_row, n_cols)
    a, b = HipShardEngine(enc), HipShardEngine(enc)
    a.comm_init()                                   # the communicator belongs to the encoder: both engines use it
    polys = [O.random_elems(fid, n_rows * n_per_row, 700 + i) for i in range(4)]
    devs = [torch.from_numpy(p.view(np.int64)).cuda() for p in polys]
    ocs = [O.Commit.commit(p, oenc, n_threads=4) for p in polys]
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for rep in range(3):
            assert a.commit_native(devs[0], n_rows, want_root=False, async_tail=True) is None
            assert b.commit_native(devs[1], n_rows, want_root=False, async_tail=True) is None
            assert a.commit_native(devs[2], n_rows, want_root=False, async_tail=True) is None      # refill: behind a's own tail
            assert b.commit_native(devs[3], n_rows, want_root=False, async_tail=True) is None
        # readers need no synchronisation either: they wait for the commitment's event
        assert a.cm.get_root() == ocs[2].get_root() and b.cm.get_root() == ocs[3].get_root()
        assert b.commit_native(devs[1], n_rows, want_root=True, async_tail=True) == ocs[1].get_root()
    st.synchronize()
    a.cm._refresh(); b.cm._refresh()
    assert (a.cm.hashes() == ocs[2].hashes()).all() and (b.cm.hashes() == ocs[1].hashes()).all()
    assert (a.cm.coeffs() == ocs[2].coeffs()).all() and (b.cm.coeffs() == ocs[1].coeffs()).all()
    outer = O.random_elems(fid, n_rows, 710)
    n_open = enc.get_n_col_opens()
    for eng, oc in ((a, ocs[2]), (b, ocs[1])):
        root = oc.get_root()
        opf, _ = oc.prove(outer, oenc, mk_transcript(O.Transcript, root, n_open))
        data, _ = eng.prove_native(outer, mk_transcript(Transcript, root, n_open))
        assert data == opf
