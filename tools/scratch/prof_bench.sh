cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_k -o k -- python3 $R/bench.py --no-cpu --steps 10 > $R/gpurun_out/h_bench_under_rocprof.json 2>/dev/null
cp /tmp/p_k/k_kernel_stats.csv $R/gpurun_out/h_kernel_stats.csv
