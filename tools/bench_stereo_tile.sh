# stereo configurations: larger launches (YDORB_BENCH_STEREO_TILE = copies of the distinct stream per launch); results in gpurun_out/
B="python bench.py --repeats 6 --no-ba --no-cpu"
YDORB_BENCH_STEREO_TILE=4 $B --only config3 > gpurun_out/r3_tile_c3_4.json 2> gpurun_out/r3_tile_c3_4.err; echo c3
YDORB_BENCH_STEREO_TILE=16 $B --only config4 > gpurun_out/r3_tile_c4_16.json 2> gpurun_out/r3_tile_c4_16.err; echo c4
