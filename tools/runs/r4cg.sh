set -e
O=gpurun_out/r4cg; mkdir -p $O
python tools/range_soak.py 150 300 2>&1 | grep -v amdgpu > $O/any_shipped.txt
RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip_ab.so python tools/range_soak.py 150 300 2>&1 | grep -v amdgpu > $O/any_ab.txt
python tools/range_soak.py 150 300 ref 2>&1 | grep -v amdgpu > $O/ref_shipped.txt
RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip_ab.so RPT_NO_SIZED_KERNELS=1 python tools/range_soak.py 150 300 ref 2>&1 | grep -v amdgpu > $O/ref_ab_general.txt
cmp $O/any_shipped.txt $O/any_ab.txt && cmp $O/ref_shipped.txt $O/ref_ab_general.txt && echo "ALL IDENTICAL"
wc -l $O/*.txt | tail -1
