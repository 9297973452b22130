# config 3: what do the association (stereo replay) and the left-frame search each cost on top of the two extractions?  results in gpurun_out/
B="python bench.py --repeats 2 --no-ba --no-cpu --only config3"
for parts in stereo match none; do
  YDORB_BENCH_STEREO_PARTS=$parts $B > gpurun_out/r3_parts_$parts.json 2> gpurun_out/r3_parts_$parts.err
  echo "done $parts"
done
for sets in 5 6; do
  YDORB_BENCH_STEREO_SETS=$sets $B > gpurun_out/r3_sets_$sets.json 2> gpurun_out/r3_sets_$sets.err
  echo "done sets $sets"
done
