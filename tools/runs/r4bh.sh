set -e
O=gpurun_out/r4bh; mkdir -p $O
python -m pytest tests/test_gpu_probes.py -m gpu -x -q > $O/probes.log 2>&1 || { tail -30 $O/probes.log; exit 1; }
tail -2 $O/probes.log
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
