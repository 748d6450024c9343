set -e
mkdir -p gpurun_out/r4h
python -m pytest tests -m gpu -x -q > gpurun_out/r4h/tests.log 2>&1 || { tail -40 gpurun_out/r4h/tests.log; exit 1; }
tail -3 gpurun_out/r4h/tests.log
python tools/host_path_time.py > gpurun_out/r4h/host_path.txt 2>&1
for o in 0 1; do for dm in 3072 0; do
  RPT_DISPATCH_ORDER=$o RPT_COMPACT_DENSE_MAX=$dm python tools/compact_time.py 800 600 1 400 >> gpurun_out/r4h/compact.txt 2>&1
done; RPT_DISPATCH_ORDER=$o python tools/compact_time.py 1920 1080 1 200 >> gpurun_out/r4h/compact.txt 2>&1; done
python tools/block_profile.py 256 c2 > gpurun_out/r4h/block_profile_c2.txt 2>&1
python tools/block_profile.py 64 c4 > gpurun_out/r4h/block_profile_c4.txt 2>&1
python tools/block_profile.py 32 c5 > gpurun_out/r4h/block_profile_c5.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4h/host_path.txt gpurun_out/r4h/compact.txt gpurun_out/r4h/block_profile_c2.txt
