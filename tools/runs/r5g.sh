O=gpurun_out/r5g; mkdir -p $O
for i in 1 2; do bash tools/run_variants.sh tools/ab_time.py c2 6; done 2>&1 | tee $O/c2.txt
bash tools/run_variants.sh tools/ab_time.py c5 4 2>&1 | tee $O/c5.txt
