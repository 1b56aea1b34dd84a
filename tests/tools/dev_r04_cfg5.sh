# development aid: the cfg5 loop with the library under test, default deadlines
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --config cfg5 --steps 400 --warmup 10 > gpurun_out/r04_cfg5_now.json 2> gpurun_out/r04_cfg5_now.err || { tail -5 gpurun_out/r04_cfg5_now.err; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04_cfg5_now.json').read().strip().splitlines()[-1])
print('value', d['value'], d['unit'], 'ms/step', d['ms_per_step'])
c=d.get('config',{}); print({k:c[k] for k in c if 'budget' in k})
x=d.get('extra',{})
print(json.dumps(x, indent=0)[:2500])
P
