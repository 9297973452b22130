# config 3, same box, alternating: 4 lanes with in-lane copies against 8 hardware queues, 7 lanes + upload stream; results in gpurun_out/
B="python bench.py --repeats 10 --no-ba --no-cpu --only config3"
for i in 1 2; do
  $B > gpurun_out/r3_c3_lane_$i.json 2> gpurun_out/r3_c3_lane_$i.err; echo "lane $i"
  GPU_MAX_HW_QUEUES=8 YDORB_BENCH_STEREO_SETS=8 YDORB_BENCH_COPY=upload $B > gpurun_out/r3_c3_up8_$i.json 2> gpurun_out/r3_c3_up8_$i.err; echo "up8 $i"
done
