# round 6: the whole GPU suite and the bench line on the current tree
set -e
O=gpurun_out/r6h; mkdir -p $O
export AMD_LOG_LEVEL=1
python -m pytest tests -m gpu -x -q --capture=sys > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
unset AMD_LOG_LEVEL
python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6h/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'frac', d['roofline']['frac'], 'scaling', d['scaling'])
print('secondary', json.dumps(d.get('secondary')))
print('six', json.dumps(d.get('six_primitives')))
print('f64', json.dumps(d.get('f64_reference')))
print('tiles', d['c3_rank_tiles']['tile_ms'], d['c3_rank_tiles']['projected_scaling_vs_whole_frame'])
PY
