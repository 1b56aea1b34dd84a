# development aid (round 4): the arm's parity tests, then timings, of the library in the tree
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${1:-cfg4 or panda or chain or arm or stale or migration or budget or fused or lane}" > gpurun_out/r04_arm_tests.log 2>&1 || { tail -40 gpurun_out/r04_arm_tests.log; exit 1; }
tail -3 gpurun_out/r04_arm_tests.log
timeout -k 10 300 python tests/tools/quick_time.py cfg4 > gpurun_out/r04_arm_qt.log 2>&1 || { tail gpurun_out/r04_arm_qt.log; exit 1; }
cat gpurun_out/r04_arm_qt.log
