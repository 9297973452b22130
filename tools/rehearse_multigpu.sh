# Rehearsals of bench.py's N > 1 code paths on a one-GPU box (results in gpurun_out/):
#  1. RCCL process group at world size 1 (in-place all-gather, reductions, barrier; with YDORB_BENCH_VARIANTS also the neighbour form)
#  2. two ranks on one GPU over gloo (collectives staged through host memory): the default all-gather form as `value`, the ring and
#     neighbour forms as `exchange.variants` of the same run
set -e
YDORB_BENCH_FORCE_DIST=1 YDORB_BENCH_VARIANTS=1 YDORB_BENCH_LANES=6 python bench.py --repeats 2 --steps 3 --no-cpu --no-extras > gpurun_out/rehearse_rccl_world1.json 2> gpurun_out/rehearse_rccl_world1.err
echo "rccl world 1 ok"
YDORB_BENCH_BACKEND=gloo YDORB_BENCH_ONE_GPU=1 python bench.py --gpus 2 --repeats 2 --steps 2 --substeps 2 --frames 128 --ba-reps 1 --cpu-frames 16 > gpurun_out/rehearse_gloo2.json 2> gpurun_out/rehearse_gloo2.err
echo "gloo 2 ranks ok"
