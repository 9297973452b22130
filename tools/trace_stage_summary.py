"""Average duration of the full-batch launches (N frames per launch) of every extractor / matcher kernel in a rocprofv3 --kernel-trace CSV,
per launch geometry: the stats file averages all launches of a kernel name, whatever their batch size (single frames, stereo configs, ...).
Usage: python tools/trace_stage_summary.py <kernel_trace.csv> [frames per launch = 512]"""
import csv, sys, collections
N = sys.argv[2] if len(sys.argv) > 2 else "512"
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void ydorb::", "").replace("ydorb::", "").split("(")[0]
    full = N in (r["Grid_Size_Y"], r["Grid_Size_Z"]) or (name.startswith("k_qt_fast") and r["Grid_Size_X"] == str(int(N) * 256)) \
        or (name in ("k_resolve", "k_grid_build") and r["Grid_Size_X"] in (str((int(N) - 1) * 64), str(int(N) * 256))) \
        or (name in ("k_gather_projection", "k_queries_from_keypoints") and r["Grid_Size_Y"] == str(int(N) - 1))
    if full and name.startswith("k_"):
        agg[(name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-28s %-22s %8s %10s %10s %10s" % ("kernel", "grid (threads)", "launches", "avg us", "min us", "max us"))
for (name, gx, gy, gz), v in sorted(agg.items()):
    print("%-28s %-22s %8d %10.1f %10.1f %10.1f" % (name, "%s,%s,%s" % (gx, gy, gz), len(v), sum(v) / len(v), min(v), max(v)))
