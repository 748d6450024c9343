set -e
O=gpurun_out/r6c; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/microbench/gather_lane_cost.hip -o /tmp/gather_lane_cost
timeout -k 10 120 /tmp/gather_lane_cost | tee $O/gather_lane_cost.txt
