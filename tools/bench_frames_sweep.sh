# frames per launch with the uploads on their own stream; results in gpurun_out/
B="python bench.py --repeats 4 --no-ba --no-cpu --only none --no-extras"
$B --frames 512 --substeps 32 > gpurun_out/r3_fr_512.json 2> gpurun_out/r3_fr_512.err; echo 512
$B --frames 1024 --substeps 16 > gpurun_out/r3_fr_1024.json 2> gpurun_out/r3_fr_1024.err; echo 1024
$B --frames 2048 --substeps 8 > gpurun_out/r3_fr_2048.json 2> gpurun_out/r3_fr_2048.err; echo 2048
