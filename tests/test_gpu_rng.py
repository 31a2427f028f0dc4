"""lcpc_random_coeffs_device: lcpc_test_fields::random_coeffs (lcpc-test-fields/src/lib.rs:75-97) with SURVEY.md 8(d)'s fixed
generator -- ChaCha20Rng::from_seed(seed), Field::random's rejection rule -- computed on the device (count / scan / place over the
candidate stream) must give, element for element, the vector the CPU oracle draws serially (oracle/lcpc_oracle.c
lo_rng_field_random).  bench.py times THIS vector and its cpu_baseline leg commits the oracle's copy of it."""
import numpy as np
import pytest

import oracle_lib as OL
from lcpc_amd import LcCommit, LigeroEncoding

pytestmark = pytest.mark.gpu


def oracle_stream(fid, n, seed, stream_id):
    L = OL.limbs(fid)
    key = np.full(32, seed & 0xFF, np.uint8)
    g = OL.lib().lo_rng_from_seed(OL.ptr(key))
    OL.lib().lo_rng_set_stream(g, stream_id)
    out = np.zeros((n, L), np.uint64)
    OL.lib().lo_rng_field_random(g, fid, OL.ptr(out), n)
    OL.lib().lo_rng_free(g)
    return out


@pytest.mark.parametrize("fid", [0, 1, 2, 3])
@pytest.mark.parametrize("n,seed,stream_id", [(1, 0, 0), (7, 1, 0), (255, 0, 0), (256, 0, 0), (1000, 7, 0), (100003, 0, 0), ((1 << 20) + 5, 3, 0),
                                              (4097, 0, 5), (70001, 9, (1 << 40) + 3)])
def test_device_random_coeffs_equal_oracle(oracle, fid, n, seed, stream_id):
    enc = LigeroEncoding.new(fid, 1 << 12)
    dev = enc.random_coeffs_device(n, seed=seed, stream_id=stream_id)
    got = dev.cpu().numpy().view(np.uint64)
    exp = oracle_stream(fid, n, seed, stream_id)
    if stream_id == 0:
        assert (exp == oracle.random_elems(fid, n, seed)).all()
    assert got.shape == exp.shape
    bad = np.nonzero((got != exp).any(axis=1))[0]
    assert bad.size == 0, (fid, n, "first mismatch at element %d" % bad[0])
    # every element is fully reduced (< p) and the stream is not degenerate
    import pyref as P
    p = P.FIELDS[fid].p
    top = got[:, -1].astype(object)
    assert int(top.max()) <= p >> (64 * fid)
    if n >= 1000:
        assert len({bytes(r) for r in got[:1000]}) == 1000


def test_device_random_coeffs_commit_root(oracle):
    """the whole point: a commitment of the device-drawn vector equals the oracle's commitment of its own copy of it"""
    O, fid, n = oracle, 3, (1 << 18) - 3
    enc = LigeroEncoding.new(fid, n)
    dev = enc.random_coeffs_device(n, seed=0)
    c = LcCommit.commit_device(dev.data_ptr(), n, enc)
    oc = O.Commit.commit(O.random_elems(fid, n, 0), O.Encoding.ligero(fid, n), n_threads=8)
    assert c.get_root() == oc.get_root() and (c.hashes() == oc.hashes()).all()
