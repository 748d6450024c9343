set -e
mkdir -p gpurun_out/r4bo
for cfg in c4 c5; do
  echo "== $cfg" >> gpurun_out/r4bo/x.txt
  bash tools/run_variants.sh tools/ab_time.py $cfg 6 >> gpurun_out/r4bo/x.txt 2>&1
done
cat gpurun_out/r4bo/x.txt
