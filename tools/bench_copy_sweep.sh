# sweep of the mono pipeline's lane count / handle stream mode (bench.py's YDORB_BENCH_* knobs); results in gpurun_out/
set -e
B="python bench.py --repeats 2 --no-ba --no-cpu --only none"
for cfg in "3 1" "4 1" "5 1" "6 1" "4 0"; do
  set -- $cfg
  YDORB_BENCH_LANES=$1 YDORB_BENCH_SINGLE_STREAM=$2 $B > gpurun_out/r3_sw2_$1_$2.json 2> gpurun_out/r3_sw2_$1_$2.err
  echo "done $cfg"
done
