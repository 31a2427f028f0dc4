"""tools/lab/ab_shapes.py -- two builds of the library against each other on one box, interleaved: a child process per (library,
shape); tools/lab/base/liblcpc_hip.so is the baseline build (copy one there), lcpc_amd/lib/liblcpc_hip.so the current one."""
import os, sys, subprocess
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "tools", "lab"))
import ab_shapes_child
child = ab_shapes_child.CHILD % R
libs = {"base": os.path.join(R, "tools/lab/base/liblcpc_hip.so"), "new ": os.path.join(R, "lcpc_amd/lib/liblcpc_hip.so")}
for log_cols, rows in ((18, 512), (17, 256), (19, 128), (20, 128), (16, 512), (22, 16)):
    for rep in range(2):
        for name, path in libs.items():
            out = subprocess.run([sys.executable, "-c", child, path, str(log_cols), str(rows)], capture_output=True, text=True)
            print("n_cols 2^%d x %d rows" % (log_cols, rows), name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
