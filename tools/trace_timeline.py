"""Print which hardware queue / stream every kernel of a rocprofv3 --kernel-trace CSV ran on, and the timeline of one bench step."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].replace('void ydorb::', '').replace('ydorb::', '').split('(')[0][:28]
q = collections.defaultdict(collections.Counter)
for r in rows: q[(r['Queue_Id'], r['Stream_Id'])][nm(r)] += 1
for k, v in q.items(): print(k, dict(v))
idx = [i for i, r in enumerate(rows) if 'k_pyr_level0' in r['Kernel_Name'] and int(r['Grid_Size_Z']) >= 128]
s, e = idx[nth], idx[min(nth + (int(sys.argv[3]) if len(sys.argv) > 3 else 1), len(idx) - 1)]
t0 = int(rows[s]['Start_Timestamp'])
for r in rows[s:e]:
    print('%-30s q=%s st=%s start=%8.1f end=%8.1f dur=%7.1f' % (nm(r), r['Queue_Id'], r['Stream_Id'], (int(r['Start_Timestamp']) - t0) / 1e3,
          (int(r['End_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
