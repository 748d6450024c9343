# round 6: the material table by class of accepted set (5-8 primitives): parity (table vs no table vs oracle, the kernel that ran), then what it buys
set -e
O=gpurun_out/r6g; mkdir -p $O
export AMD_LOG_LEVEL=1
python -m pytest tests/test_gpu_dispatch.py -m gpu -x -q --capture=sys > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python tools/table_shape_time.py 2>&1 | tee $O/table_shape_time.txt
