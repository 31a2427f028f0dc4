"""tools/lab/ab_field_libs.py fid log_len name=path ... -- several builds on one Ligero commit of another field, interleaved (child of ab_fields.py)"""
import os, sys, subprocess, re
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
src = open(os.path.join(R, "tools/lab/ab_fields.py")).read()
child = src[src.index("child = r'''") + len("child = r'''"):src.index("''' % R")] % R
fid, lg = sys.argv[1], sys.argv[2]
libs = [a.split("=", 1) for a in sys.argv[3:]]
for rep in range(3):
    for name, path in libs:
        out = subprocess.run([sys.executable, "-c", child, os.path.join(R, path), fid, lg], capture_output=True, text=True)
        print("field %s 2^%s %-6s" % (fid, lg, name), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
