"""mean of each PMC counter per kernel name (last third of dispatches) from rocprofv3 --pmc csv output"""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('void aslam::', '')[:40]
    by[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in by.items():
    if 'large' not in k and 'small' not in k: continue
    print(k)
    for c, v in cs.items():
        v = v[-max(1, len(v) // 3):]
        print('    %-28s %.4g' % (c, sum(v) / len(v)))
