import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tools"))
import bench_configs as B
from lcpc_amd import LigeroEncoding
for rho in ((1, 2), (1, 4), (3, 4)):
    B.time_commit("ligero ft255 2^26 rho=%d/%d" % rho, LigeroEncoding.new(3, 1 << 26, rho), 1 << 26, 4)
