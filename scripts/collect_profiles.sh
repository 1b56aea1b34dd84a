#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + two PMC passes of
# the default bench workload.  Outputs under gpurun_out/; summarise afterwards with
# scripts/summarize_profiles.py and commit the summaries under profiles/.
# Counters are collected in their own runs (no trace domains besides --kernel-trace).
set -e
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r01}
CFG=${2:-cfg2}
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench_$TAG.json 2> gpurun_out/prof_bench_$TAG.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -- python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2> gpurun_out/pmc_fetch_$TAG.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$TAG -- python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2> gpurun_out/pmc_write_$TAG.err
find gpurun_out -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | sort
