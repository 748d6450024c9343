# configs[4]'s grid knobs on the kernel that shares samples (round 5)
O=gpurun_out/r5_sweep; mkdir -p $O; : > $O/sweep_grid.txt
for spc in 0.5 0.75 1 1.5 2; do RPT_GRID_SPHERES_PER_CELL=$spc timeout -k 10 200 python tools/ab_time.py c5 4 2>&1 | grep -v amdgpu.ids >> $O/sweep_grid.txt || exit 1; done
for nr in 0 1 1.5 2 3; do RPT_GRID_NEAR_REACH=$nr timeout -k 10 200 python tools/ab_time.py c5 4 2>&1 | grep -v amdgpu.ids >> $O/sweep_grid.txt || exit 1; done
for st in 48 56; do for um in 16 32 64; do RPT_SHADE_THRESHOLD=$st RPT_UNIT_MIN_SPP=$um timeout -k 10 200 python tools/ab_time.py c5l 3 2>&1 | grep -v amdgpu.ids >> $O/sweep_grid.txt || exit 1; done; done
cut -c1-40,60-220 $O/sweep_grid.txt
