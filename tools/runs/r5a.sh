# round 5, batch a: this box's baselines, the finer block profile of configs[4], the denoiser's counters and XCD-banded tiles,
# the FMA-issue microbenchmark with the clock counted.
O=gpurun_out/r5a; mkdir -p $O
for c in c2 c4 c5; do python tools/ab_time.py $c 5 2>/dev/null | tail -1 >> $O/base.txt; done
python tools/ab_time.py c5full 2 2>/dev/null | tail -1 >> $O/base.txt
cat $O/base.txt
python tools/block_profile.py 32 c5 > $O/block_profile_c5.txt 2>&1; tail -32 $O/block_profile_c5.txt
bash tools/run_variants.sh tools/ab_time.py dn 10 > $O/dn_variants.txt 2>&1; cat $O/dn_variants.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_issue_peak.hip -o /tmp/valu_issue_peak && /tmp/valu_issue_peak > $O/valu_issue_peak.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/rp_peak -o out -- /tmp/valu_issue_peak > /dev/null 2>&1
cp $(find /tmp/rp_peak -name '*counter_collection.csv' | head -1) $R/$O/valu_issue_peak_pmc.csv
cp $(find /tmp/rp_peak -name '*kernel_trace.csv' | head -1) $R/$O/valu_issue_peak_trace.csv
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/rp_dn1 -o out -- python3 $R/tools/ab_time.py dn 3 > /dev/null 2>&1
cp $(find /tmp/rp_dn1 -name '*counter_collection.csv' | head -1) $R/$O/dn_pmc1.csv
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU --output-format csv -d /tmp/rp_dn2 -o out -- python3 $R/tools/ab_time.py dn 3 > /dev/null 2>&1
cp $(find /tmp/rp_dn2 -name '*counter_collection.csv' | head -1) $R/$O/dn_pmc2.csv
cd $R; cat $O/valu_issue_peak.txt; ls -la $O
