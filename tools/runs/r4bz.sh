set -e
O=gpurun_out/r4bz; mkdir -p $O
python tools/range_soak.py 60 1 ref 2>&1 | grep -v amdgpu > $O/ref_shipped.txt
RPT_NO_MATERIAL_TABLE=1 python tools/range_soak.py 60 1 ref 2>&1 | grep -v amdgpu > $O/ref_no_table.txt
RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip_ab.so python tools/range_soak.py 60 1 ref 2>&1 | grep -v amdgpu > $O/ref_ab.txt
RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip_ab.so RPT_NO_SIZED_KERNELS=1 python tools/range_soak.py 60 1 ref 2>&1 | grep -v amdgpu > $O/ref_ab_general.txt
python tools/range_soak.py 60 200 2>&1 | grep -v amdgpu > $O/any_shipped.txt
RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip_ab.so python tools/range_soak.py 60 200 2>&1 | grep -v amdgpu > $O/any_ab.txt
cmp $O/ref_shipped.txt $O/ref_no_table.txt && cmp $O/ref_shipped.txt $O/ref_ab.txt && cmp $O/ref_shipped.txt $O/ref_ab_general.txt && cmp $O/any_shipped.txt $O/any_ab.txt && echo "ALL IDENTICAL"
wc -l $O/*.txt
