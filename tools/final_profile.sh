# End-of-round evidence on the GPU box (from the repo root): bash tools/final_profile.sh <tag>
#   gpurun_out/<tag>_bench.json                  default `python bench.py` (un-profiled)
#   gpurun_out/<tag>_trace/...                   rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu --steps 3 --repeats 2`
#   gpurun_out/<tag>_bench_2048frame_launches.txt  per-launch durations of the 2048-frame launches of that trace
#   gpurun_out/pmc_<tag>/...                     four separate --pmc passes over the extractor alone (tools/pmc_extract.sh)
#   gpurun_out/<tag>_gpu_tests.log, gpurun_out/<tag>_fuzz.log
TAG=${1:-r03}
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err && echo "bench ok"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- python3 bench.py --no-cpu --steps 3 --repeats 2 > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_bench_under_rocprof.err && echo "trace ok"
python3 tools/trace_stage_summary.py $(ls gpurun_out/${TAG}_trace/*/*kernel_trace.csv | tail -1) 2048 > gpurun_out/${TAG}_bench_2048frame_launches.txt
bash tools/pmc_extract.sh $TAG 256 > gpurun_out/${TAG}_pmc.log 2>&1 && python3 tools/pmc_to_bench_csv.py gpurun_out/pmc_$TAG 256 gpurun_out/$TAG >> gpurun_out/${TAG}_pmc.log 2>&1 && echo "pmc ok"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.log 2>&1; tail -2 gpurun_out/${TAG}_gpu_tests.log
timeout -k 10 420 python tools/fuzz_parity.py 300 20261005 > gpurun_out/${TAG}_fuzz.log 2>&1; tail -1 gpurun_out/${TAG}_fuzz.log
