set -e
mkdir -p gpurun_out/r4j
python -m pytest tests -m gpu -x -q -k "sdf or dispatch" > gpurun_out/r4j/tests.log 2>&1 || { tail -40 gpurun_out/r4j/tests.log; exit 1; }
tail -2 gpurun_out/r4j/tests.log
for i in 1 2; do
bash tools/run_variants.sh tools/ab_time.py c4 8 >> gpurun_out/r4j/c4.txt 2>&1
bash tools/run_variants.sh tools/ab_time.py c5 8 >> gpurun_out/r4j/c5.txt 2>&1
done
cat gpurun_out/r4j/c4.txt gpurun_out/r4j/c5.txt
