#!/bin/bash
# tools/pmc_tile_group.sh -- FETCH_SIZE / WRITE_SIZE of the K1s first pass at 2^20 columns (128 rows, owning commit) with the
# grouped tile order (default) and the plain XCD-aware order (LCPC_NTT_TILE_GROUP=0); separate --pmc passes, kernel trace only.
# Run on the MI355X box:  gpurun -- 'bash tools/pmc_tile_group.sh > gpurun_out/tile_group_pmc.txt'
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/tg_commit.py <<'P'
import sys
sys.path.insert(0, sys.argv[1])
import torch
from lcpc_amd import LcCommit, LigeroEncoding
log_n, rows = 20, 128
n_cols, npr = 1 << log_n, 1 << (log_n - 1)
g = torch.Generator(device="cuda"); g.manual_seed(1)
coeffs = torch.randint(0, 1 << 62, (rows * npr, 4), dtype=torch.int64, device="cuda", generator=g)
coeffs[:, 3] &= (1 << 60) - 1
enc = LigeroEncoding.new_from_dims(3, npr, n_cols)
c = LcCommit(enc)
for _ in range(3):
    LcCommit.commit_device(coeffs.data_ptr(), rows * npr, enc, torch.cuda.current_stream().cuda_stream, sync=True, into=c)
P
db() { find $1 -name '*.db' | head -1; }
for tg in default 0; do
  if [ $tg = default ]; then unset LCPC_NTT_TILE_GROUP; else export LCPC_NTT_TILE_GROUP=$tg; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tg_$ctr
    rocprofv3 --pmc $ctr -d /tmp/tg_$ctr -o x -- python /tmp/tg_commit.py $R > /tmp/tg.log 2>&1
    echo "== LCPC_NTT_TILE_GROUP=$tg $ctr (KB per dispatch; FETCH_SIZE counts half the bytes of wide reads on gfx950)"
    python $R/tools/rocpd_summary.py $(db /tmp/tg_$ctr) --pmc | grep -E "KERNEL|ntt_pass_l9s"
  done
done
