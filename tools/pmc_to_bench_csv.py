"""Turn the PMC passes of tools/pmc_extract.sh into the two small CSVs bench.py reads for roofline.traffic / valu_busy_frac.
Usage: python tools/pmc_to_bench_csv.py gpurun_out/pmc_<tag> <frames per launch> profiles/<tag>
FETCH_SIZE is calibrated on k_pyr_level0 of the same pass, which must read W*H bytes per frame (640x480 here): the counter
under-reports this 4-byte access pattern (MI355X_MICROARCH.md, HBM section: only 16-B-per-lane streams have a published factor)."""
import csv, glob, os, sys, collections
root, frames, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
def last(name, counter):
    d = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(root, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ydorb::", "").split("<")[0]
                d[k][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
    res = {}
    for k, v in d.items():
        ids = sorted(v)
        # launches of the last batch: a stage may have several launches per batch (k_pyr_resize x7, k_qt_fast per level group)
        n_per_batch = max(1, len(ids) // 6)
        res[k] = sum(v[i] for i in ids[-n_per_batch:])
    return res
fe, wr = last("fetch", "FETCH_SIZE"), last("write", "WRITE_SIZE")
lvl0 = "k_pyr_level0_f" if "k_pyr_level0_f" in fe else "k_pyr_level0"   # the level-0 kernel of the plan in use
corr = (640 * 480 * frames) / (fe[lvl0] * 1024.0)
with open(out + "_pmc_hbm_traffic.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "fetch_MB_per_frame_raw", "write_MB_per_frame", "fetch_correction", "frames_per_launch"])
    for k in sorted(fe):
        if k.startswith("k_"):
            w.writerow([k, "%.4f" % (fe[k] * 1024 / 1e6 / frames), "%.4f" % (wr.get(k, 0) * 1024 / 1e6 / frames), "%.3f" % corr, frames])
sq = {c: last("sq", c) for c in ("SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_WAVES")}
with open(out + "_pmc_sq_valu.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "VALU_insts_per_wave", "VALU_busy_pct_of_SIMD_cycles"])
    for k in sorted(sq["SQ_WAVES"]):
        if k.startswith("k_") and sq["GRBM_GUI_ACTIVE"].get(k):
            w.writerow([k, "%.1f" % (sq["SQ_INSTS_VALU"][k] / max(sq["SQ_WAVES"][k], 1)),
                        "%.1f" % (100 * 4 * sq["SQ_ACTIVE_INST_VALU"][k] / (1024 * sq["GRBM_GUI_ACTIVE"][k] / 8))])
print("fetch correction", corr)
