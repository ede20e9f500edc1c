"""mean of each PMC counter per kernel name from `rocprofv3 --pmc ... --output-format csv` output, over the dispatches with
grids of at least a quarter of the kernel's largest (the full-batch launches, all block columns of the large path:
bench.py also runs a batch-1 leg) -- the last half of those.
Also prints the derived MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES): BUSY_CYCLES accumulates per
shader engine (32 of them, 32 SIMDs each), MFMA_BUSY per SIMD."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
grid = collections.defaultdict(int)
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('void aslam::', '')[:40]
    grid[k] = max(grid[k], int(r['Grid_Size']))
by = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('void aslam::', '')[:40]
    if 4 * int(r['Grid_Size']) >= grid[k]:
        by[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in by.items():
    if 'large' not in k and 'small' not in k and 'scan' not in k: continue
    m = {c: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for c, v in cs.items()}
    print(k, '(grid %d threads, %d dispatches)' % (grid[k], len(next(iter(cs.values())))))
    for c, v in m.items():
        print('    %-28s %.4g' % (c, v))
    if m.get('SQ_BUSY_CYCLES') and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        print('    -> MFMA utilisation           %.1f %%' % (100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (32 * m['SQ_BUSY_CYCLES'])))
    if m.get('SQ_WAVE_CYCLES'):
        print('    -> of wave time: waiting (waitcnt/barrier) %.0f %%, issue-stalled %.0f %%, issuing %.0f %%' % tuple(
            100 * m.get(c, 0) / m['SQ_WAVE_CYCLES'] for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY')))
