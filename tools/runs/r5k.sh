O=gpurun_out/r5k; mkdir -p $O
bash tools/run_variants.sh tools/ab_time.py c2 5 2>&1 | tee $O/waves.txt
for st in 48 52 56 60 64; do for ft in 16 24 32; do RPT_SHADE_THRESHOLD=$st RPT_FINISH_THRESHOLD=$ft python tools/ab_time.py c2 4 2>/dev/null | tail -1; done; done | tee $O/thresholds.txt
