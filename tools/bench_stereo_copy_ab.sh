# stereo configurations: uploads on the last lane's stream (default) against uploads in every lane; results in gpurun_out/
B="python bench.py --repeats 4 --no-ba --no-cpu --only config3,config4"
$B > gpurun_out/r3_sc_upload.json 2> gpurun_out/r3_sc_upload.err && echo "upload ok"
YDORB_BENCH_COPY=lane YDORB_BENCH_UPLOAD_STREAM=0 $B > gpurun_out/r3_sc_lane.json 2> gpurun_out/r3_sc_lane.err && echo "lane ok"
