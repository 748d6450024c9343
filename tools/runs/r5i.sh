O=gpurun_out/r5i; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_denoise.py tests/test_shipped_library.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for v in dn_fused2 dn_unfused; do RPT_LIB=$PWD/rust-pathtracer_amd/variants/$v.so python tools/ab_time.py dn 10 2>/dev/null | grep "^dn"; done | tee $O/dn.txt
