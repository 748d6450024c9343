set -e
mkdir -p gpurun_out/r4am
RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip_ab.so python -m pytest tests -m gpu -x -q > gpurun_out/r4am/tests_ab.log 2>&1 || { tail -60 gpurun_out/r4am/tests_ab.log; exit 1; }
tail -3 gpurun_out/r4am/tests_ab.log
python bench.py --steps 20 --warmup 3 > gpurun_out/r4am/bench.json 2> gpurun_out/r4am/bench.err || { tail -30 gpurun_out/r4am/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4am/bench.json') if l.startswith('{')][-1])
print(d['value'], d['roofline']['frac'], d['roofline']['ieee_expanded_frac'], d['roofline_c4']['value'], d['roofline_c5']['value'], d.get('config1'))
PY
