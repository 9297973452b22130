# sweep of the inclusive pipeline's copy placement / lane count / handle stream mode (bench.py's YDORB_BENCH_* knobs); results in gpurun_out/
set -e
B="python bench.py --repeats 2 --no-ba --no-cpu --only none"
for cfg in "lane 5 1" "lane 6 1" "lane 8 1" "streams 4 1"; do
  set -- $cfg
  YDORB_BENCH_COPY=$1 YDORB_BENCH_LANES=$2 YDORB_BENCH_SINGLE_STREAM=$3 $B > gpurun_out/r3_sw_$1_$2_$3.json 2> gpurun_out/r3_sw_$1_$2_$3.err
  echo "done $cfg"
done
