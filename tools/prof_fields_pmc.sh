#!/bin/bash
# tools/prof_fields_pmc.sh -- SQ counters of the small-field NTT passes (tools/bench_fields.py, Ligero only): where K1n's time goes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_fields
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
  -d $O/sq -o f -- python $R/tools/bench_fields.py 24 ligero > $O/sq.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/sq -name '*.db' | head -1) --pmc | grep -E "ntt_pass|leaf_chunk|KERNEL" > $O/fields_pmc_sq.txt
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE \
  -d $O/sq2 -o f -- python $R/tools/bench_fields.py 24 ligero > $O/sq2.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/sq2 -name '*.db' | head -1) --pmc | grep -E "ntt_pass|leaf_chunk|KERNEL" > $O/fields_pmc_sq2.txt
rm -rf $O/sq $O/sq2
cat $O/fields_pmc_sq.txt $O/fields_pmc_sq2.txt
