#!/bin/bash
# tools/lab/pmc_icache.sh TAG -- instruction-cache and scalar-cache counters of the two headline NTT kernels (own --pmc passes)
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_icache_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ \
  -d $O/a -o bench -- python $R/bench.py --steps 2 --warmup 1 --lean > $O/a.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/a -name '*.db' | head -1) --pmc | grep -E "ntt_pass|leaf_chunk" > $O/icache.txt
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
  -d $O/b -o bench -- python $R/bench.py --steps 2 --warmup 1 --lean > $O/b.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/b -name '*.db' | head -1) --pmc | grep -E "ntt_pass|leaf_chunk" >> $O/icache.txt
rm -rf $O/a $O/b
cat $O/icache.txt
