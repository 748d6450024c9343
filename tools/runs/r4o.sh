set -e
mkdir -p gpurun_out/r4o
python -m pytest tests -m gpu -x -q > gpurun_out/r4o/tests.log 2>&1 || { tail -40 gpurun_out/r4o/tests.log; exit 1; }
tail -2 gpurun_out/r4o/tests.log
RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip_ab.so python -m pytest tests -m gpu -x -q > gpurun_out/r4o/tests_ab.log 2>&1 || { tail -40 gpurun_out/r4o/tests_ab.log; exit 1; }
tail -2 gpurun_out/r4o/tests_ab.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r4o/bench.json 2> gpurun_out/r4o/bench.err || { tail -20 gpurun_out/r4o/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4o/bench.json"))
print("value", d["value"], "frac", d["roofline"]["frac"])
for k in ("roofline_c4","roofline_c5"):
    print(k, d[k]["value"], d[k]["frac"], d[k].get("progressive_8spp"))
PY
