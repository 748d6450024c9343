O=gpurun_out/r5_soak; mkdir -p $O
for v in shipped perop; do RPT_LIB=$PWD/rust-pathtracer_amd/variants/$v.so python tools/range_soak.py 80 500 2>/dev/null | grep "^scene" > $O/$v.txt; RPT_LIB=$PWD/rust-pathtracer_amd/variants/$v.so python tools/range_soak.py 40 900 ref 2>/dev/null | grep "^scene" > $O/${v}_ref.txt; done
RPT_NO_SIZED_KERNELS=1 RPT_LIB=$PWD/rust-pathtracer_amd/variants/shipped.so python tools/range_soak.py 40 900 ref 2>/dev/null | grep "^scene" > $O/shipped_ref_general.txt
cmp $O/shipped.txt $O/perop.txt && cmp $O/shipped_ref.txt $O/perop_ref.txt && cmp $O/shipped_ref.txt $O/shipped_ref_general.txt && echo "SOAK: 80 + 40 + 40 scenes byte-identical across builds" | tee $O/result.txt
wc -l $O/*.txt; grep -c "nan pixels     0" $O/shipped.txt
