"""tools/lab/ab_quick.py -- the headline commit (2^18 columns x 512 rows) on two builds of the library, interleaved, three rounds:
tools/lab/base/liblcpc_hip.so against lcpc_amd/lib/liblcpc_hip.so (child process per run; ab_shapes.py has the other shapes)."""
import os, sys, subprocess
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "tools", "lab"))
import ab_shapes_child as ch
libs = {"base": os.path.join(R, "tools/lab/base/liblcpc_hip.so"), "new ": os.path.join(R, "lcpc_amd/lib/liblcpc_hip.so")}
for rep in range(3):
    for name, path in libs.items():
        out = subprocess.run([sys.executable, "-c", ch.CHILD % R, path, "18", "512"], capture_output=True, text=True)
        print("2^18 x 512", name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
