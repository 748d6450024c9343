O=gpurun_out/r5_class_costs; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/microbench/cndmask_costs.hip -o /tmp/cndmask_costs && timeout -k 5 90 /tmp/cndmask_costs > $O/cndmask.txt; echo rc=$?; cat $O/cndmask.txt
