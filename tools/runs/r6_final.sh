# round 6, last: what the driver runs at round end — smoke(), the GPU suite (-x), the default bench line
set -e
O=gpurun_out/r6_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt
python -m pytest tests/ -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6_final/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'scaling', d['scaling'])
print('valu_issue', json.dumps(d['roofline'].get('valu_issue',{}).get('insts_per_simd_quad_cycle')))
print('secondary', json.dumps(d.get('secondary')))
print('c4 traffic', d['roofline_c4']['traffic'], 'c5 traffic', d['roofline_c5']['traffic'], 'dn4k', d['roofline_denoise_4k'].get('counter_GBs'), d['roofline_denoise_4k'].get('wait_inst_lds_share_of_wave_time'))
print('f64', d['f64_reference']['gpu_f32_vs_f64']['rmse'], d['f64_reference']['gpu_f32_bit_identical_to_strict_oracle'])
PY
