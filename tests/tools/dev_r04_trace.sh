# development aid (round 4): rocprofv3 kernel trace of several handles in flight, summarised with tests/tools/trace_load.py
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
rm -rf gpurun_out/r04_trace4
QT_STREAMS=4 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04_trace4 -- python3 tests/tools/quick_time.py cfg4 > gpurun_out/r04_trace4.log 2> gpurun_out/r04_trace4.err
cat gpurun_out/r04_trace4.log
f=$(find gpurun_out/r04_trace4 -name "*kernel_trace.csv" | head -1)
python3 tests/tools/trace_load.py $f 20
