set -e
mkdir -p gpurun_out/r4ba
python -m pytest tests/test_gpu_range_guards.py -m gpu -x -q > gpurun_out/r4ba/tests.log 2>&1 || { tail -40 gpurun_out/r4ba/tests.log; exit 1; }
tail -2 gpurun_out/r4ba/tests.log
python tools/range_soak.py 60 100 2>/dev/null | grep scene > gpurun_out/r4ba/shipped.txt
RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip_ab.so python tools/range_soak.py 60 100 2>/dev/null | grep scene > gpurun_out/r4ba/ab.txt
if diff gpurun_out/r4ba/shipped.txt gpurun_out/r4ba/ab.txt > gpurun_out/r4ba/diff.txt; then echo "60 more scenes: identical"; else echo DIFFERENT; head -20 gpurun_out/r4ba/diff.txt; fi
awk '{print $9, $10}' gpurun_out/r4ba/shipped.txt | sort | uniq -c | sort -rn | head -5
