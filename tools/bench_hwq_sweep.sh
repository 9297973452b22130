# stereo configurations: hardware queues per process (GPU_MAX_HW_QUEUES, default 4) x lanes x copy placement; results in gpurun_out/
B="python bench.py --repeats 3 --no-ba --no-cpu --only config3,config4"
for cfg in "4 4 lane" "4 4 upload" "8 6 upload" "8 8 upload" "8 8 lane"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 YDORB_BENCH_STEREO_SETS=$2 YDORB_BENCH_COPY=$3 $B > gpurun_out/r3_hwq_$1_$2_$3.json 2> gpurun_out/r3_hwq_$1_$2_$3.err
  echo "done $cfg"
done
