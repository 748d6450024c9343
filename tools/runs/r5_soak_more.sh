# 400 more scenes (other seeds) through the final kernels: the shipped library against the per-operation build, and the general kernels
O=gpurun_out/r5_soak_more; mkdir -p $O
for v in shipped perop; do RPT_LIB=$PWD/rust-pathtracer_amd/variants/$v.so python tools/range_soak.py 300 3000 2>/dev/null | grep "^scene" > $O/$v.txt; done
RPT_NO_SIZED_KERNELS=1 RPT_NO_MATERIAL_TABLE=1 RPT_LIB=$PWD/rust-pathtracer_amd/variants/shipped.so python tools/range_soak.py 100 3000 2>/dev/null | grep "^scene" > $O/shipped_general.txt
head -100 $O/shipped.txt > $O/shipped_first100.txt
cmp $O/shipped.txt $O/perop.txt && cmp $O/shipped_first100.txt $O/shipped_general.txt && echo "SOAK: 300 scenes byte-identical between the shipped library and the per-operation build, the first 100 also through the general kernels" | tee $O/result.txt
wc -l $O/*.txt
