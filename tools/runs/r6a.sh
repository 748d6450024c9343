# round 6, first contact: the GPU suite on the changed tree (configs[2] whole through 8 ranks, ragged SDF waves), A/B of the SDF kernel, the bench line
set -e
O=gpurun_out/r6a; mkdir -p $O
python -m pytest tests -m gpu -x -q --capture=sys > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for k in 1 2; do python tools/ab_time.py c4 5; done > $O/ab_c4.txt 2>&1; cat $O/ab_c4.txt
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6a/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'frac', d['roofline']['frac'], 'scaling', d['scaling'])
print('secondary', json.dumps(d.get('secondary')))
print('c3_rank_tiles', json.dumps(d.get('c3_rank_tiles')))
print('cpu', json.dumps(d.get('cpu_baseline')))
print('dn4k', json.dumps(d.get('roofline_denoise_4k')))
PY
