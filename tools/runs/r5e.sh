O=gpurun_out/r5e; mkdir -p $O
( time python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5e/bench.json') if l.startswith('{')][-1])
print('value',d['value'],'ms',d['ms_per_step'])
for k in ('general_kernels','relaxed'):
    print(k, json.dumps(d.get(k))[:700])
print('valu_issue', json.dumps(d['roofline'].get('valu_issue')), d['roofline'].get('traffic'), d['roofline'].get('traffic_source'))
print('c4',d['roofline_c4']['value'],'c5',d['roofline_c5']['value'], 'dn', d['roofline_denoise_4k']['achieved'])
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
timeout -k 10 900 python -m pytest tests/test_gpu_multi.py -q -x > $O/multi.log 2>&1; echo "multi rc=$?"; tail -3 $O/multi.log
