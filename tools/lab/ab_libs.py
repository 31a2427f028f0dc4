"""tools/lab/ab_libs.py name=path ... [-- log_cols rows] -- several builds of the library on one box, interleaved, three rounds (child per run)"""
import os, sys, subprocess
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "tools", "lab"))
import ab_shapes_child as ch
args = sys.argv[1:]
shape = ["18", "512"]
if "--" in args:
    i = args.index("--"); shape = args[i + 1:i + 3]; args = args[:i]
libs = [a.split("=", 1) for a in args]
for rep in range(3):
    for name, path in libs:
        out = subprocess.run([sys.executable, "-c", ch.CHILD % R, os.path.join(R, path), shape[0], shape[1]], capture_output=True, text=True)
        print("2^%s x %s %-6s" % (shape[0], shape[1], name), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
