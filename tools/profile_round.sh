#!/bin/bash
# tools/profile_round.sh TAG -- everything profiles/ holds for one state of the code, as text, under gpurun_out/prof_TAG/:
#   bench_line.json           python bench.py (un-profiled, with the CPU baseline and the configs; run AFTER the counter passes, which it quotes)
#   bench_kernel_stats.txt    rocprofv3 --kernel-trace --stats of the same command (no CPU baseline)
#   bench_pmc_{fetch,write,sq}.txt + pmc.json     separate --pmc passes (never combined with traces)
#   clock_power.txt           rocm-smi sclk / socket power sampled while the bench loops
#   configs.jsonl, c3_kernel_stats.txt, c3_dispatches.txt, c3_pmc.json, c5_kernel_stats.txt, c4_rank.jsonl, fields.jsonl, pvs.jsonl,
#   bench_rs.jsonl, small_trace.txt   the other BASELINE.json configs, fields, the reference's loops and cargo-bench matrix, small commits
# Run on the MI355X box:  gpurun -- 'bash tools/profile_round.sh r01f'
TAG=${1:-latest}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 5 --warmup 1 --lean"
db() { find $1 -name '*.db' | head -1; }
rocprofv3 --kernel-trace --stats -d $O/kt -o bench -- $BENCH > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(db $O/kt) > $O/bench_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o bench -- python $R/bench.py --steps 2 --warmup 1 --lean > $O/fetch.log 2>&1
python $R/tools/rocpd_summary.py $(db $O/fetch) --pmc > $O/bench_pmc_fetch.txt
rocprofv3 --pmc WRITE_SIZE -d $O/write -o bench -- python $R/bench.py --steps 2 --warmup 1 --lean > $O/write.log 2>&1
python $R/tools/rocpd_summary.py $(db $O/write) --pmc > $O/bench_pmc_write.txt
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
  -d $O/sq -o bench -- python $R/bench.py --steps 2 --warmup 1 --lean > $O/sq.log 2>&1
python $R/tools/rocpd_summary.py $(db $O/sq) --pmc > $O/bench_pmc_sq.txt
python $R/tools/pmc_to_json.py $(db $O/fetch) $(db $O/write) $O/pmc.json $(db $O/sq) > /dev/null      # -> copy to profiles/pmc_latest.json
# the bench line LAST among the bench.py runs, with this call's counters and instruction mix in place (on this box's copy of the
# repo), so that roofline.traffic / valu_issue of bench_line.json speak about the very sources being run
cp $O/pmc.json $R/profiles/pmc_latest.json
python $R/tools/isa_mix.py $R/profiles/valu_mix_latest.json > /dev/null 2> $O/isa_mix.log && cp $R/profiles/valu_mix_latest.json $O/valu_mix.json
python $R/bench.py > $O/bench_line.json 2> $O/bench_stderr.log
# clocks / power while the bench loops
python $R/bench.py --steps 1200 --warmup 2 --no-cpu-baseline > $O/clk_bench.log 2>&1 &
BP=$!
echo "# rocm-smi samples every 0.5 s during: python bench.py --steps 1200 (idle samples dropped): sclk, socket power (W)" > $O/clock_power.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | sed -e 's/.*sclk clock level: //' -e 's/.*Power (W)://' | tr '\n' ' '
  echo
  sleep 0.5
done | grep -v "([0-9][0-9]Mhz)\|([0-9][0-9][0-9]Mhz)" >> $O/clock_power.txt
tail -1 $O/clk_bench.log | cut -c1-220 >> $O/clock_power.txt
# the other configs
python $R/tools/bench_configs.py 2> $O/configs.log | grep "^{" > $O/configs.jsonl     # (RCCL prints its version banner on stdout)
rocprofv3 --kernel-trace --stats -d $O/c3 -o c3 -- python $R/tools/bench_configs.py c3 > $O/c3.log 2>&1
python $R/tools/rocpd_summary.py $(db $O/c3) > $O/c3_kernel_stats.txt
python $R/tools/rocpd_dispatches.py $(db $O/c3) | tail -24 > $O/c3_dispatches.txt      # the last commit, launch by launch
# HBM traffic and VALU counters of the Brakedown commit's kernels (separate --pmc passes, never combined with traces)
rocprofv3 --pmc FETCH_SIZE -d $O/c3f -o c3 -- python $R/tools/bench_configs.py c3 > $O/c3f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/c3w -o c3 -- python $R/tools/bench_configs.py c3 > $O/c3w.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/c3s -o c3 -- python $R/tools/bench_configs.py c3 > $O/c3s.log 2>&1
python $R/tools/pmc_to_json.py $(db $O/c3f) $(db $O/c3w) $O/c3_pmc.json $(db $O/c3s) > /dev/null
rm -rf $O/c3f $O/c3w $O/c3s
python $R/tools/bench_c4_rank.py > $O/c4_rank.jsonl 2> $O/c4_rank.log
python $R/tools/bench_fields.py 24 > $O/fields.jsonl 2> $O/fields.log                   # all four test fields, Ligero and Brakedown, 2^24
python $R/tools/bench_pvs.py > $O/pvs.jsonl 2> $O/pvs.log                              # the reference's rough_bench / prove_verify_size_bench loops
rocprofv3 --kernel-trace --stats -d $O/c5 -o c5 -- python $R/tools/bench_configs.py c5 > $O/c5.log 2>&1
python $R/tools/rocpd_summary.py $(db $O/c5) > $O/c5_kernel_stats.txt
python $R/tools/bench_rs.py 2> $O/bench_rs.log | grep "^{" > $O/bench_rs.jsonl            # the reference's cargo-bench matrix (Ft127 / Ft255 x 2^16 / 2^20 / 2^24)
bash $R/tools/trace_small.sh > $O/small_trace.txt 2>&1                                   # small commitments, launch by launch
rm -rf $O/kt $O/fetch $O/write $O/sq $O/c3 $O/c5     # raw .db files stay out of the merge-back
ls -la $O
