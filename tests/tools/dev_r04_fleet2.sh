# development aid (round 4): the warm-start and scene GPU tests in one call
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_warm_start.py tests/test_gpu_scene.py -x -q -m gpu > gpurun_out/r04_fleet_tests.log 2>&1 || { tail -30 gpurun_out/r04_fleet_tests.log; exit 1; }
tail -3 gpurun_out/r04_fleet_tests.log
