set -e
mkdir -p gpurun_out/r4v
for i in 1 2; do for c in c2 c4 c5; do bash tools/run_variants.sh tools/ab_time.py $c 8 >> gpurun_out/r4v/ab.txt 2>&1; done; done
cat gpurun_out/r4v/ab.txt
