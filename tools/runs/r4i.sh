set -e
mkdir -p gpurun_out/r4i
python -m pytest tests -m gpu -x -q > gpurun_out/r4i/tests.log 2>&1 || { tail -40 gpurun_out/r4i/tests.log; exit 1; }
tail -3 gpurun_out/r4i/tests.log
RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip_ab.so python -m pytest tests -m gpu -x -q > gpurun_out/r4i/tests_ab.log 2>&1 || { tail -40 gpurun_out/r4i/tests_ab.log; exit 1; }
tail -3 gpurun_out/r4i/tests_ab.log
python tools/host_path_time.py > gpurun_out/r4i/host_path.txt 2>&1
python tools/compact_time.py 800 600 1 400 >> gpurun_out/r4i/compact.txt 2>&1
python tools/compact_time.py 1920 1080 1 200 >> gpurun_out/r4i/compact.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4i/host_path.txt gpurun_out/r4i/compact.txt
