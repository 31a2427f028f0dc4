#!/bin/bash
# tools/lab/pmc_lds.sh TAG -- the LDS counters of the two headline NTT kernels (separate --pmc pass, nothing else traced)
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_lds_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE \
  -d $O/sq -o bench -- python $R/bench.py --steps 2 --warmup 1 --lean > $O/sq.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/sq -name '*.db' | head -1) --pmc | grep "ntt_pass" > $O/lds.txt
rm -rf $O/sq
cat $O/lds.txt
