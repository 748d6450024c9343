R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM SQ_WAIT_ANY -d $R/gpurun_out/r1e_pmcA_c5 -o run --output-format csv -- python3 $R/tools/c4_run.py c5 > $R/gpurun_out/r1e_A.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES -d $R/gpurun_out/r1e_pmcB_c5 -o run --output-format csv -- python3 $R/tools/c4_run.py c5 > $R/gpurun_out/r1e_B.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/r1e_pmcA_c5/run_counter_collection.csv render
python3 $R/tools/pmc_summary.py $R/gpurun_out/r1e_pmcB_c5/run_counter_collection.csv render
