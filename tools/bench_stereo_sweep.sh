# sweep of the stereo configurations' launch size / lane count (bench.py's YDORB_BENCH_STEREO_* knobs); results in gpurun_out/
set -e
for cfg in "1 4" "2 4" "4 4" "2 3" "4 3"; do
  set -- $cfg
  YDORB_BENCH_STEREO_TILE=$1 YDORB_BENCH_STEREO_SETS=$2 python bench.py --repeats 2 --steps 2 --substeps 4 --no-ba --no-cpu --only config3,config4 > gpurun_out/r3_st_$1_$2.json 2> gpurun_out/r3_st_$1_$2.err
  echo "done $cfg"
done
