O=gpurun_out/r5c; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "grid_queries or large or config5 or tiers" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
bash tools/run_variants.sh tools/ab_time.py c5 5 > $O/variants.txt 2>&1; cat $O/variants.txt
