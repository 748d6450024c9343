set -e
mkdir -p gpurun_out/r4x
for i in 1 2; do bash tools/run_variants.sh tools/ab_time.py dn 20 >> gpurun_out/r4x/dn.txt 2>&1; done
cat gpurun_out/r4x/dn.txt
