"""K1s against the shelved lane = row kernel K1u (tools/lab/README.md): run as a script on the GPU box, never collected by pytest."""
import os, sys, time
sys.path[:0] = [os.environ.get("GRAFT_REPO_ROOT", "/root/repo")]
import numpy as np, torch
from lcpc_amd import LcCommit, LigeroEncoding


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    npr, nc = 1 << 17, 1 << 18
    n = rows * npr - 5
    os.environ["LCPC_NTT_U"] = "0"
    enc_s = LigeroEncoding.new_from_dims(3, npr, nc)
    del os.environ["LCPC_NTT_U"]
    enc_u = LigeroEncoding.new_from_dims(3, npr, nc)
    dev = enc_u.random_coeffs_device(n, seed=3)
    a = LcCommit.commit_device(dev.data_ptr(), n, enc_s)
    b = LcCommit.commit_device(dev.data_ptr(), n, enc_u)
    print("roots equal:", a.get_root() == b.get_root())
    ca, cb = a.comm(0, min(rows, 2)), b.comm(0, min(rows, 2))
    bad = np.nonzero((ca != cb).any(axis=1))[0]
    print("row 0-1 mismatches:", bad.size, bad[:16])
    if bad.size == 0 and a.get_root() != b.get_root():
        for r in range(0, rows, 7):
            x, y = a.comm(r, 1), b.comm(r, 1)
            bb = np.nonzero((x != y).any(axis=1))[0]
            if bb.size: print("row", r, "mismatches", bb.size, bb[:8]); break
    print("coeffs equal:", (a.coeffs() == b.coeffs()).all())
    st = torch.cuda.current_stream().cuda_stream
    for enc, name in ((enc_s, "K1s"), (enc_u, "K1u")):
        c = LcCommit(enc); c.set_timing(True)
        for _ in range(3): LcCommit.commit_device(dev.data_ptr(), n, enc, st, into=c)
        t = c.timings(); print(name, "encode_ms %.3f hash %.3f total %.3f" % (t.encode_ms, t.hash_ms, t.total_ms))
    P = 0x663c799b6e4d2900fda9df04b9575969ef73c79086595f3002a4f20000000001
    def val(e): return sum(int(e[i]) << (64 * i) for i in range(4))
    badrows = []
    for r in range(rows):
        x, y = a.comm(r, 1), b.comm(r, 1)
        bb = np.nonzero((x != y).any(axis=1))[0]
        if bb.size: badrows.append((r, bb.size, int(bb[0]), int(bb[-1])))
    print("bad rows:", badrows[:70])
    if badrows:
        r, _, p0, _ = badrows[0]
        x, y = a.comm(r, 1), b.comm(r, 1)
        for pos in (p0, p0 + 1, p0 + 5):
            vx, vy = val(x[pos]), val(y[pos])
            print(pos, hex(vx)[:20], hex(vy)[:20], "diff mod p:", (vx - vy) % P == 0, "ratio-ish", (vy * pow(vx, -1, P)) % P if vx else None)


if __name__ == "__main__":
    main()
