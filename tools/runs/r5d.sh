O=gpurun_out/r5d; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "grid_queries or large or config5 or tiers or light" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for c in c5 c5full; do python tools/ab_time.py $c 4 2>/dev/null | tail -1; done | tee $O/times.txt
