set -e
O=gpurun_out/r6q; mkdir -p $O
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6q/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'steps', d['steps'])
print(json.dumps(d['f64_reference'])[:900])
PY
