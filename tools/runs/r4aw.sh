set -e
mkdir -p gpurun_out/r4aw
python -m pytest tests/test_gpu_multi.py -m gpu -x -q > gpurun_out/r4aw/tests.log 2>&1 || { tail -60 gpurun_out/r4aw/tests.log; exit 1; }
tail -3 gpurun_out/r4aw/tests.log
python -c "import __graft_entry__ as e; e.smoke(); print('smoke ok')" > gpurun_out/r4aw/smoke.log 2>&1 || { tail -20 gpurun_out/r4aw/smoke.log; exit 1; }
tail -2 gpurun_out/r4aw/smoke.log
