# A/B of library variants: bash tools/runs/r5_ab.sh "<variants>" "<workloads>" [reps]
O=gpurun_out/r5_ab; mkdir -p $O; : > $O/ab.txt
for w in $2; do
  for pass in 1 2; do
    for v in $1; do
      RPT_LIB=$PWD/rust-pathtracer_amd/variants/$v.so timeout -k 10 150 python tools/ab_time.py $w ${3:-5} 2>&1 | grep -v amdgpu.ids >> $O/ab.txt || exit 1
    done
  done
done
cat $O/ab.txt
