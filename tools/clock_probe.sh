# does the extract-only rate depend on the lane count, and what do the clocks do under the sustained load?  results in gpurun_out/
B="python bench.py --repeats 3 --no-ba --no-cpu --only none --no-extras"
( for i in $(seq 1 150); do
    echo "t=$i $(cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | grep '\*' | tr '\n' ' ') $(cat /sys/class/drm/card*/device/hwmon/hwmon*/power1_average 2>/dev/null | tr '\n' ' ')"
    sleep 0.5
  done ) > gpurun_out/r3_clk.log 2>&1 &
CLK=$!
for lanes in 1 2 4; do
  YDORB_BENCH_LANES=$lanes $B > gpurun_out/r3_lanes_$lanes.json 2> gpurun_out/r3_lanes_$lanes.err
  echo "done lanes $lanes"
done
kill $CLK 2>/dev/null
python tools/bench_stage.py 512 > gpurun_out/r3_stage_burst.txt 2>&1
YDORB_STAGE_REPS=400 python tools/bench_stage.py 512 > gpurun_out/r3_stage_sustained.txt 2>&1
echo "probe done"
