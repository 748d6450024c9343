set -e
mkdir -p gpurun_out/r4aj
for cfg in c2 c4 c5; do
  echo "== $cfg" >> gpurun_out/r4aj/x.txt
  bash tools/run_variants.sh tools/ab_time.py $cfg 6 >> gpurun_out/r4aj/x.txt 2>&1
done
cat gpurun_out/r4aj/x.txt
RPT_LIB=$PWD/rust-pathtracer_amd/variants/base.so python -m pytest tests -m gpu -x -q -k "range_guards or probe or math or config2 or c2" > gpurun_out/r4aj/tests.log 2>&1 || { tail -60 gpurun_out/r4aj/tests.log; exit 1; }
tail -3 gpurun_out/r4aj/tests.log
