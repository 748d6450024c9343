set -e
mkdir -p gpurun_out/r4bk
bash tools/run_variants.sh tools/ab_time.py c4 6 >> gpurun_out/r4bk/x.txt 2>&1
cat gpurun_out/r4bk/x.txt
