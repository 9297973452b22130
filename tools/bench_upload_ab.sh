# same box, alternating: 4 lanes with in-lane uploads against 3 lanes + upload stream (image ROCm runtime); results in gpurun_out/
B="python bench.py --repeats 6 --no-ba --no-cpu --only none --no-extras"
for i in 1 2; do
  YDORB_BENCH_LANES=4 YDORB_BENCH_UPLOAD_STREAM=0 $B > gpurun_out/r3_ab_lane_$i.json 2> gpurun_out/r3_ab_lane_$i.err; echo "lane $i"
  YDORB_BENCH_LANES=3 YDORB_BENCH_UPLOAD_STREAM=1 $B > gpurun_out/r3_ab_up_$i.json 2> gpurun_out/r3_ab_up_$i.err; echo "up $i"
done
