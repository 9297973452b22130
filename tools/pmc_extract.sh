#!/bin/bash
# PMC passes over the extractor only (tools/bench_stage.py F frames per launch): SQ pass, LDS pass, FETCH pass, WRITE pass.
# Usage (on the GPU box, from the repo root): bash tools/pmc_extract.sh <tag> [frames]
TAG=${1:-r02}; F=${2:-256}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { name=$1; shift; echo "pass $name"; timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 tools/bench_stage.py $F > $OUT/$name.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE
run lds SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
