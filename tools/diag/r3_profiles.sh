#!/bin/bash
# Round-3 measurement passes on the GPU box (run through gpurun from the repo root):
#   bash tools/diag/r3_profiles.sh stats|traffic|sq
# Every rocprofv3 pass is a run of its own (PMC passes never share a run with --stats / other counters' blocks).
# Summaries: tools/diag/r3_summaries.py (run in the build container afterwards) -> profiles/r3_*.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# one sampling chain per pass (DM_SPLIT=1): a launch then covers the whole batch, as in bench.py's per-class pass whose launch
# times these summaries are compared with; the default (four sub-batch chains on parallel graph branches)
# overlaps launches of a quarter of the batch each, whose durations say nothing about a kernel on its own.  `stats4` below
# traces the default once for the record.
export DM_SPLIT=1
case "$1" in
stats)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > $O/prof_cfg2.json 2> $O/prof_cfg2.err || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg1 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs --sampler ancestral --ddim-steps 1000 --batch 64 > $O/prof_cfg1.json 2> $O/prof_cfg1.err || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --precision bf16 --cfg > $O/prof_cfg4.json 2> $O/prof_cfg4.err || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --model wide-attn --batch 256 > $O/prof_cfg5.json 2> $O/prof_cfg5.err || exit 1
  ;;
stats4)
  unset DM_SPLIT
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg2_split4 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-roofline > $O/prof_cfg2_split4.json 2> $O/prof_cfg2_split4.err || exit 1
  ;;
traffic)
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-configs > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 1
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-configs > $O/pmc_write.json 2> $O/pmc_write.err || exit 1
  ;;
sq)
  # SQ counters over one DDIM-50 pass (B=1024): MFMA-busy fraction and vector instructions per MFMA per kernel
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_model -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-configs > $O/pmc_sq_model.json 2> $O/pmc_sq_model.err || exit 1
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_cfg1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-configs --sampler ancestral --ddim-steps 100 --batch 64 > $O/pmc_sq_cfg1.json 2> $O/pmc_sq_cfg1.err || exit 1
  ;;
sq4)
  # the same counters over one bf16 + guidance pass (BASELINE configs[3])
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_cfg4 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-configs --precision bf16 --cfg > $O/pmc_sq_cfg4.json 2> $O/pmc_sq_cfg4.err || exit 1
  ;;
esac
echo done $1
