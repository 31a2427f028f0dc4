R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lab_pmc; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
db() { find $1 -name '*.db' | head -1; }
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d $O/a -o lab -- $R/tools/ntt_lab 9 brief > $O/a.log 2>&1
python $R/tools/rocpd_summary.py $(db $O/a) --pmc > $O/pmc_a.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d $O/b -o lab -- $R/tools/ntt_lab 9 brief > $O/b.log 2>&1
python $R/tools/rocpd_summary.py $(db $O/b) --pmc > $O/pmc_b.txt 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU -d $O/c -o lab -- $R/tools/ntt_lab 9 brief > $O/c.log 2>&1
python $R/tools/rocpd_summary.py $(db $O/c) --pmc > $O/pmc_c.txt 2>&1
rm -rf $O/a $O/b $O/c
tail -3 $O/a.log
