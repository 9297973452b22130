# inclusive pipeline with the uploads on one stream of their own (YDORB_BENCH_UPLOAD_STREAM=1) against the in-lane form; results in gpurun_out/
set -e
B="python bench.py --repeats 3 --no-ba --no-cpu --only none --no-extras"
for cfg in "4 0" "3 1" "2 1" "4 1"; do
  set -- $cfg
  YDORB_BENCH_LANES=$1 YDORB_BENCH_UPLOAD_STREAM=$2 $B > gpurun_out/r3_up_$1_$2.json 2> gpurun_out/r3_up_$1_$2.err
  echo "done $cfg"
done
