# A/B: the select pass of build.py (VOP2 -> VOP3 encodings of v_cndmask_b32) on the four workloads
O=gpurun_out/r5_select; mkdir -p $O; : > $O/ab.txt
for w in c2 c4 c5 c2s; do
  for v in sel_none sel_runs sel_all sel_none sel_runs sel_all; do
    RPT_LIB=$PWD/rust-pathtracer_amd/variants/$v.so timeout -k 10 120 python tools/ab_time.py $w 5 >> $O/ab.txt 2>&1 || exit 1
  done
done
cat $O/ab.txt
