O=gpurun_out/r5_valu_peak; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/microbench/valu_issue_peak.hip -o /tmp/valu_issue_peak && /tmp/valu_issue_peak > $O/events.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/rp_peak -o out -- /tmp/valu_issue_peak > /dev/null 2>&1
cp $(find /tmp/rp_peak -name '*counter_collection.csv' | head -1) $R/$O/pmc.csv
cp $(find /tmp/rp_peak -name '*kernel_trace.csv' | head -1) $R/$O/trace.csv
cd $R; python tools/valu_peak_summary.py $O/pmc.csv $O/trace.csv | tee $O/summary.txt
