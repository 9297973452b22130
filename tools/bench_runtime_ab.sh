# bench.py on the image's ROCm runtime (default) against the runtime bundled with PyTorch (YDORB_BENCH_SYSTEM_ROCM=0); results in gpurun_out/
B="python bench.py --repeats 5 --no-ba --no-cpu --only config3,config4"
$B > gpurun_out/r3_rt_image.json 2> gpurun_out/r3_rt_image.err && echo "image runtime ok"
YDORB_BENCH_SYSTEM_ROCM=0 $B > gpurun_out/r3_rt_torch.json 2> gpurun_out/r3_rt_torch.err && echo "torch runtime ok"
