cd /root/repo
mkdir -p gpurun_out/r04i
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r04i/bench.json 2> gpurun_out/r04i/bench.err ) 2> gpurun_out/r04i/time.txt
tail -c 600 gpurun_out/r04i/bench.err; cat gpurun_out/r04i/time.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04i/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','kernel_ms','roofline','roofline_valu','binding_bound')})
print(d['cpu_baseline'])
for k,v in d['extras'].items():
    if isinstance(v,dict): print(k, {kk:v.get(kk) for kk in ('kernel_ms','kernel','roofline_frac','bound','parity_bit_exact_on_sample','traffic')})
    else: print(k,v)
PY
