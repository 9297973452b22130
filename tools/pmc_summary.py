"""Summarise the rocprofv3 --pmc passes written by tools/pmc_extract.sh: per kernel, the counters of the LAST batch of launches
(averaged per dispatch), plus derived figures."""
import csv, glob, os, sys, collections
root = sys.argv[1]
def load(name):
    rows = []
    for f in glob.glob(os.path.join(root, name, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows
def short(k):
    k = k.split("(")[0]
    return k.replace("void ", "").replace("ydorb::", "")[:40]
def agg(rows):
    # kernel -> counter -> list of values per dispatch (ordered)
    d = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in rows:
        d[short(r["Kernel_Name"])][r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
    out = {}
    for k, cs in d.items():
        out[k] = {}
        for c, byid in cs.items():
            ids = sorted(byid)
            last = ids[len(ids) * 4 // 5:] or ids      # the last fifth of the dispatches (steady state)
            out[k][c] = sum(byid[i] for i in last) / len(last)
        out[k]["_n"] = len(next(iter(cs.values())))
    return out
sq, lds, fe, wr = agg(load("sq")), agg(load("lds")), agg(load("fetch")), agg(load("write"))
print("%-40s %8s %10s %10s %8s %8s %8s %8s %9s %9s" % ("kernel", "disp", "valu/wave", "lds/wave", "VALUbusy", "LDSbusy", "waitAny", "waitInst", "fetchKB", "writeKB"))
for k in sorted(sq):
    s = sq[k]; l = lds.get(k, {})
    waves = max(s.get("SQ_WAVES", 1), 1)
    simd_cyc = 1024 * s.get("GRBM_GUI_ACTIVE", 0) / 8      # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    valu_busy = 4 * s.get("SQ_ACTIVE_INST_VALU", 0) / simd_cyc if simd_cyc else 0
    lds_busy = 4 * l.get("SQ_ACTIVE_INST_LDS", 0) / simd_cyc if simd_cyc else 0
    wc = max(s.get("SQ_WAVE_CYCLES", 1), 1)
    print("%-40s %8d %10.1f %10.1f %8.3f %8.3f %8.3f %8.3f %9.1f %9.1f" % (k, s["_n"], s.get("SQ_INSTS_VALU", 0) / waves, l.get("SQ_INSTS_LDS", 0) / max(l.get("SQ_WAVES", waves), 1) if "SQ_WAVES" in l else l.get("SQ_INSTS_LDS", 0) / waves,
          valu_busy, lds_busy, s.get("SQ_WAIT_ANY", 0) / wc, s.get("SQ_WAIT_INST_ANY", 0) / wc, fe.get(k, {}).get("FETCH_SIZE", 0), wr.get(k, {}).get("WRITE_SIZE", 0)))
