#!/bin/bash
# Development aid (run on the GPU box): solo and overlapped throughput of the three robot families.
for cfg in cfg2 cfg3 cfg4; do
  python3 bench.py --config $cfg --streams 1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg solo', round(d['value']), round(d['ms_per_step'],2), d['solve_stats']['passes_last_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['roofline']['all_kernels'].items()})"
  python3 bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-events 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg x4  ', round(d['value']), round(d['ms_per_step'],2))"
done
