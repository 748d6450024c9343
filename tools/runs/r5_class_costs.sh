O=gpurun_out/r5_class_costs; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/microbench/valu_class_costs.hip -o /tmp/valu_class_costs && timeout -k 5 90 /tmp/valu_class_costs > $O/events.txt; echo rc=$?; cat $O/events.txt
