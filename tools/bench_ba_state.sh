# what in the bench process slows the single BA solve's host side?  results in gpurun_out/
B="python bench.py --repeats 2 --no-cpu --no-extras"
$B > gpurun_out/r3_bas_default.json 2> gpurun_out/r3_bas_default.err; echo a
$B --frames 1024 --substeps 16 > gpurun_out/r3_bas_f1024.json 2> gpurun_out/r3_bas_f1024.err; echo b
YDORB_BENCH_SYSTEM_ROCM=0 $B > gpurun_out/r3_bas_torchrt.json 2> gpurun_out/r3_bas_torchrt.err; echo c
GPU_MAX_HW_QUEUES=4 $B > gpurun_out/r3_bas_q4.json 2> gpurun_out/r3_bas_q4.err; echo d
