"""Sum a PMC counter (FETCH_SIZE / WRITE_SIZE, in KB) over the dispatches of the LAST bench step of a
`rocprofv3 --pmc X --kernel-trace -- python3 bench.py --workload W --steps K ...` run.
usage: pmc_traffic.py <rocprof output dir> <kernel-name substring> <dispatches per bench step>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Dispatch_Id']))
per = int(sys.argv[3])
last = rows[-per:]
name = last[0]['Counter_Name']
print(name, 'KB over the last', len(last), 'dispatches matching', repr(sys.argv[2]), '=', sum(float(r['Counter_Value']) for r in last))
