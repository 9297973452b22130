#!/bin/bash
# One rocprofv3 counter pass over the extractor alone: bash tools/pmc_pass.sh <name> <frames> COUNTER...   (on the GPU box, from the repo root)
NAME=$1; F=$2; shift 2
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$NAME
mkdir -p $OUT
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 tools/bench_stage.py $F > $OUT.log 2>&1
python3 - "$OUT" <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not f: print('no counter file'); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'].replace('void ydorb::', '').replace('ydorb::', '').split('(')[0][:26] + ' g' + r['Grid_Size']
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print('%-44s' % k, ' '.join('%s=%.4g' % (c, sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:]))) for c, v in sorted(d.items())))
P
