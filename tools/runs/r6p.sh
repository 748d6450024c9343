O=gpurun_out/r6p; mkdir -p $O
for c in c4 c5; do RPT_DISPATCH_TIMELINE=1 python tools/dispatch_timeline.py $c 2>&1 | grep -v amdgpu.ids | tee $O/timeline_$c.txt; done
