set -e
O=gpurun_out/final_check; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python -c "
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1])
print('default bench:', d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['valu_issue']['insts_per_simd_quad_cycle'], d['roofline']['valu_issue']['frac_of_measured_ceiling'], d['roofline_c4']['value'], d['roofline_c5']['value'], d['cpu_baseline']['value'], d['gpu_over_cpu'])
"
