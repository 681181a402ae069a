#!/bin/bash
# Developer tool: collect the round's rocprofv3 artifacts on the GPU box (run from the repo root: bash tests/collect_profiles.sh TAG).
# Counters are collected in their own passes with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 section).
TAG=${1:-r03}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -o step -- python3 $R/bench.py --steps 4 --warmup 2 \
  --kernel-reps 0 --no-optimizer --no-cpu-baseline --no-secondary > $O/step_line.json 2> $O/step.err
echo "step trace rc=$?"
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
ATT_LAYOUT=head rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pmc_attn -o a -- python3 $R/tests/bench_attn.py 64 2 > $O/pmc_attn.log 2>&1
echo "pmc attn rc=$?"
ATT_LAYOUT=head rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/pmc_attn2 -o a -- python3 $R/tests/bench_attn.py 64 2 > $O/pmc_attn2.log 2>&1
echo "pmc attn2 rc=$?"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pmc_tn -o a -- python3 $R/tests/bench_one.py tn 93312 4352 1152 3 > $O/pmc_tn.log 2>&1
echo "pmc tn rc=$?"
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pmc_nt -o a -- python3 $R/tests/bench_one.py nt 93312 4352 1152 3 > $O/pmc_nt.log 2>&1
echo "pmc nt rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o a -- python3 $R/tests/bench_one.py nt 93312 4352 1152 3 > $O/pmc_fetch.log 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o a -- python3 $R/tests/bench_one.py nt 93312 4352 1152 3 > $O/pmc_write.log 2>&1
echo "pmc write rc=$?"
cd $R
python3 tests/prof_summary.py $O/step 6 30 > $O/step_kernel_stats.txt 2>&1
for d in pmc_attn pmc_attn2 pmc_tn pmc_nt pmc_fetch pmc_write; do echo "## $d"; python3 tests/pmc_summary.py $O/$d; done > $O/pmc_summary.txt 2>&1
# keep the merge-back small: drop the raw traces, keep the summaries and the per-kernel CSVs
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
du -sh $O
