#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes of the DEFAULT bench command (same steps / warm-up / spin-up
# as the driver's run, so the trace's steady-state launches are the ones bench.py times), then the steady-state summary.
# Usage: tools/prof.sh <tag> [extra bench.py args]   -> gpurun_out/prof_<tag>/...   (copy the summaries into profiles/)
set -u
TAG=${1:-r2}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --no-cpu-baseline --no-other-dtypes $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
python3 tools/trace_summary.py $OUT $OUT/$TAG > $OUT/steady.log 2>&1
PCMD="python3 bench.py --steps 20 --warmup 5 --spinup-steps 10 --no-cpu-baseline --no-other-dtypes $*"
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- $PCMD > $OUT/pmc_$N.log 2>&1
done
python3 tools/pmc_summary.py $OUT $OUT/${TAG}_pmc.json --latest $OUT/pmc_latest.json --precision ${PREC:-f16c} > $OUT/pmc_summary.log 2>&1
cat $OUT/steady.log; tail -1 $OUT/trace.log
