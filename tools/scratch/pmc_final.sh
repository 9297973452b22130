cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -o f -- python3 $R/tools/bench_stage.py 128 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -o w -- python3 $R/tools/bench_stage.py 128 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_s -o s -- python3 $R/tools/bench_stage.py 128 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_k -o k -- python3 $R/bench.py --no-cpu --steps 10 > $R/gpurun_out/f_bench_under_rocprof.json 2>/dev/null
cp /tmp/p_f/f_counter_collection.csv $R/gpurun_out/f_fetch.csv
cp /tmp/p_w/w_counter_collection.csv $R/gpurun_out/f_write.csv
cp /tmp/p_s/s_counter_collection.csv $R/gpurun_out/f_sq.csv
cp /tmp/p_k/k_kernel_stats.csv $R/gpurun_out/f_kernel_stats.csv
ls -la $R/gpurun_out/f_*
