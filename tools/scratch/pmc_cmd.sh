cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_s -o s -- python3 $GRAFT_REPO_ROOT/tools/bench_stage.py 128 > /dev/null 2>&1
cp /tmp/p_s/s_counter_collection.csv $GRAFT_REPO_ROOT/gpurun_out/e_sq.csv
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace --output-format csv -d /tmp/p_s2 -o s -- python3 $GRAFT_REPO_ROOT/tools/bench_stage.py 128 > /dev/null 2>&1
cp /tmp/p_s2/s_counter_collection.csv $GRAFT_REPO_ROOT/gpurun_out/e_sq2.csv
