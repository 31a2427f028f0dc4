"""tools/lab/ab_fields.py -- two builds of the library against each other on one box, interleaved, on the Ligero commits of the other
test fields (K1n) and on Ft255's general kernel (LCPC_NTT_GENERAL=1): a child process per (library, case);
tools/lab/base/liblcpc_hip.so is the baseline build, lcpc_amd/lib/liblcpc_hip.so the current one.  Prints root prefix, ms per commit,
encode ms."""
import os, sys, subprocess
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
child = r'''
import os, sys, time
sys.path[:0] = [%r]
import lcpc_amd._lib as L
L.LIB_PATH = sys.argv[1]
import torch
from lcpc_amd import LcCommit, LigeroEncoding
fid, lg = int(sys.argv[2]), int(sys.argv[3])
n = 1 << lg
enc = LigeroEncoding.new(fid, n)
dev = enc.random_coeffs_device(n, seed=0)
st = torch.cuda.current_stream().cuda_stream
c = LcCommit(enc)
root = LcCommit.commit_device(dev.data_ptr(), n, enc, st, into=c).get_root()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15:
    for _ in range(5): LcCommit.commit_device(dev.data_ptr(), n, enc, st, sync=False, into=c)
    torch.cuda.synchronize()
reps = 0; t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3 or reps < 20:
    for _ in range(10): LcCommit.commit_device(dev.data_ptr(), n, enc, st, sync=False, into=c)
    torch.cuda.synchronize(); reps += 10
ms = (time.perf_counter() - t0) / reps * 1e3
c.set_timing(True); LcCommit.commit_device(dev.data_ptr(), n, enc, st, into=c); t = c.timings()
print(root.hex()[:16], round(ms, 3), round(t.encode_ms, 3))
''' % R
libs = {"base": os.path.join(R, "tools/lab/base/liblcpc_hip.so"), "new ": os.path.join(R, "lcpc_amd/lib/liblcpc_hip.so")}
cases = [(0, 24, {}), (1, 24, {}), (2, 24, {}), (1, 26, {}), (2, 20, {}), (3, 24, {"LCPC_NTT_GENERAL": "1"}), (3, 18, {"LCPC_NTT_GENERAL": "1"})]
for fid, lg, env in cases:
    for rep in range(2):
        for name, path in libs.items():
            out = subprocess.run([sys.executable, "-c", child, path, str(fid), str(lg)], capture_output=True, text=True, env={**os.environ, **env})
            print("field %d 2^%d %s" % (fid, lg, " ".join("%s=%s" % kv for kv in env.items())), name,
                  out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
