"""tools/lab/ab_shapes.py -- two builds of the library against each other on one box, interleaved: a child process per (library,
shape); tools/lab/base/liblcpc_hip.so is the baseline build (copy one there), lcpc_amd/lib/liblcpc_hip.so the current one."""
import os, sys, subprocess
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
child = r'''
import os, sys, time
sys.path[:0] = [%r]
import lcpc_amd._lib as L
L.LIB_PATH = sys.argv[1]
import torch
from lcpc_amd import LcCommit, LigeroEncoding
log_cols, rows = int(sys.argv[2]), int(sys.argv[3])
npr = 1 << (log_cols - 1)
n = rows * npr
enc = LigeroEncoding.new_from_dims(3, npr, 2 * npr)
dev = enc.random_coeffs_device(n, seed=0)
st = torch.cuda.current_stream().cuda_stream
c = LcCommit(enc)
root = LcCommit.commit_device(dev.data_ptr(), n, enc, st, into=c).get_root()
for _ in range(5): LcCommit.commit_device(dev.data_ptr(), n, enc, st, sync=False, into=c)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): LcCommit.commit_device(dev.data_ptr(), n, enc, st, sync=False, into=c)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
print(root.hex()[:16], round(ms, 3))
''' % R
libs = {"base": os.path.join(R, "tools/lab/base/liblcpc_hip.so"), "new ": os.path.join(R, "lcpc_amd/lib/liblcpc_hip.so")}
for log_cols, rows in ((18, 512), (17, 256), (19, 128), (20, 128), (16, 512), (22, 16)):
    for rep in range(2):
        for name, path in libs.items():
            out = subprocess.run([sys.executable, "-c", child, path, str(log_cols), str(rows)], capture_output=True, text=True)
            print("n_cols 2^%d x %d rows" % (log_cols, rows), name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
