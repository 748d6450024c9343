set -e
mkdir -p gpurun_out/r4p
python -m pytest tests -m gpu -x -q -k "multi or tiling or gather or resident or bench" > gpurun_out/r4p/tests.log 2>&1 || { tail -60 gpurun_out/r4p/tests.log; exit 1; }
tail -3 gpurun_out/r4p/tests.log
python tools/two_stream_time.py 1920 1080 256 6 c2 2 2>&1 | grep -v amdgpu
