# knobs re-swept on the kernels that share samples (round 5)
O=gpurun_out/r5_sweep; mkdir -p $O; : > $O/sweep.txt
run() { timeout -k 10 150 python tools/ab_time.py $1 4 2>&1 | grep -v amdgpu.ids >> $O/sweep.txt || exit 1; }
for st in 48 52 56 60 64; do for ft in 16 24 32; do RPT_SHADE_THRESHOLD=$st RPT_FINISH_THRESHOLD=$ft run c2; done; done
for r in 8 12 16 24; do for m in 32 64 128; do RPT_UNIT_ROUNDS=$r RPT_UNIT_MIN_SPP=$m run c2; done; done
for mm in 2 4 8 12 16 24; do RPT_SDF_MARCH_MIN_LANES=$mm run c4; done
for st in 32 40 48 56 64; do RPT_SHADE_THRESHOLD=$st run c5; done
for r in 8 12 16 24; do for m in 16 32 64; do RPT_UNIT_ROUNDS=$r RPT_UNIT_MIN_SPP=$m run c4; done; done
cat $O/sweep.txt
