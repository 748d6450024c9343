# round 6: 300 random small scenes scaled by 2^-33 ... 2^33 (tests/scene_fuzz.py: 1-8 spheres with whole materials on the checker floor:
# 5-8 primitives take the material table by class), the shipped library with its tables against the same library building every
# material per hit: one hash per scene, the two lists must be equal
O=gpurun_out/r6_range_soak; mkdir -p $O
export RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip_test.so
python tools/range_soak.py 300 6000 2>/dev/null | grep "^scene" > $O/tables.txt
RPT_NO_MATERIAL_TABLE=1 python tools/range_soak.py 300 6000 2>/dev/null | grep "^scene" > $O/per_hit.txt
RPT_NO_MATERIAL_TABLE=1 RPT_NO_SIZED_KERNELS=1 RPT_SHADE_THRESHOLD=30 python tools/range_soak.py 100 6000 2>/dev/null | grep "^scene" > $O/per_hit_general_100.txt
head -100 $O/tables.txt > $O/tables_100.txt
wc -l $O/*.txt
cmp $O/tables.txt $O/per_hit.txt && cmp $O/tables_100.txt $O/per_hit_general_100.txt && echo "SOAK: 300 scaled scenes byte-identical with the material tables (2^n rows / by class) and with the material built per hit" | tee $O/result.txt
awk '{print $NF}' $O/tables.txt | sort | uniq -d | head -3
