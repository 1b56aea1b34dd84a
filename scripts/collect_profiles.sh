#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace (4 streams as in the timed region, and 1 stream =
# exclusive durations) + PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters; each set in its own run, no trace domain
# besides --kernel-trace) of bench.py for one config.  Outputs under gpurun_out/; summarise afterwards with
# scripts/summarize_profiles.py and commit the summaries under profiles/.
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r02}
CFG=${2:-cfg2}
B="--config $CFG --no-cpu-baseline --no-legs"
rm -rf gpurun_out/prof_${TAG}_${CFG}*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_${CFG}_s4 -- python3 bench.py $B --steps 24 --warmup 4 > gpurun_out/prof_${TAG}_${CFG}_s4.json 2> gpurun_out/prof_${TAG}_${CFG}_s4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_${CFG}_s1 -- python3 bench.py $B --steps 12 --warmup 2 --streams 1 --no-full-chip > gpurun_out/prof_${TAG}_${CFG}_s1.json 2> gpurun_out/prof_${TAG}_${CFG}_s1.err
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/prof_${TAG}_${CFG}_pmc$i -- python3 bench.py $B --steps 4 --warmup 1 --streams 1 --no-kernel-events --input-sets 2 > /dev/null 2> gpurun_out/prof_${TAG}_${CFG}_pmc$i.err || tail -2 gpurun_out/prof_${TAG}_${CFG}_pmc$i.err
done
find gpurun_out -path "*prof_${TAG}_${CFG}*" -name "*.csv" | sort | head -40
