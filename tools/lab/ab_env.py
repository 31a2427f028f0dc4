"""tools/lab/ab_env.py VAR v1 v2 ... [-- log_cols rows] -- the current library with an environment switch at several values, interleaved,
three rounds, a child process per run (the switches are read when a context is created): root prefix and ms per commit."""
import os, sys, subprocess
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "tools", "lab"))
import ab_shapes_child as ch
args = sys.argv[1:]
shape = ["18", "512"]
if "--" in args:
    i = args.index("--"); shape = args[i + 1:i + 3]; args = args[:i]
var, vals = args[0], args[1:]
lib = os.path.join(R, "lcpc_amd/lib/liblcpc_hip.so")
for rep in range(3):
    for v in vals:
        out = subprocess.run([sys.executable, "-c", ch.CHILD % R, lib, shape[0], shape[1]], capture_output=True, text=True, env={**os.environ, var: v})
        print("2^%s x %s %s=%s" % (shape[0], shape[1], var, v), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
