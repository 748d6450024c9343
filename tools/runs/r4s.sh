set -e
mkdir -p gpurun_out/r4s
for i in 1 2; do
bash tools/run_variants.sh tools/ab_time.py c2 8 >> gpurun_out/r4s/c2.txt 2>&1
bash tools/run_variants.sh tools/ab_time.py c4 8 >> gpurun_out/r4s/c4.txt 2>&1
done
cat gpurun_out/r4s/c2.txt gpurun_out/r4s/c4.txt
