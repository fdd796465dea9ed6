"""Where the GPU idles during a solve: reads a rocprofv3 --kernel-trace CSV (Start_Timestamp / End_Timestamp per dispatch) and
prints busy time, the distribution of the gaps between consecutive kernels, and the time lost in gaps above a threshold (host
round trips: chunk boundaries, top-ups).   python tools/gap_analysis.py <..._kernel_trace.csv> [first_kernel_substring]"""
import csv
import sys

import numpy as np

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
st = np.array([r[0] for r in rows], dtype=np.int64); en = np.array([r[1] for r in rows], dtype=np.int64)
names = [r[2] for r in rows]
# analyse the span from the first to the last slot kernel (the solves), skipping setup
idx = [i for i, nm in enumerate(names) if 'k_slot' in nm]
lo, hi = idx[0], idx[-1]
st, en, names = st[lo:hi + 1], en[lo:hi + 1], names[lo:hi + 1]
busy = (en - st).sum(); span = en[-1] - st[0]
gaps = st[1:] - np.maximum.accumulate(en)[:-1]
gaps = np.maximum(gaps, 0)
print('dispatches %d, span %.2f ms, busy %.2f ms (%.1f %%), idle %.2f ms' % (len(st), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
for lo_us, hi_us in ((0, 2), (2, 5), (5, 20), (20, 100), (100, 1000), (1000, 1e9)):
    sel = (gaps >= lo_us * 1e3) & (gaps < hi_us * 1e3)
    print('  gaps %6g - %6g us: %6d, total %.2f ms' % (lo_us, hi_us, sel.sum(), gaps[sel].sum() / 1e6))
# the solves alone: maximal runs of dispatches without a gap above 200 us (what lies between them is the bench's own work: set-up of the next
# step, the timing probes and their state copies), runs with fewer than 500 dispatches dropped
cuts = np.where(gaps > 200e3)[0]
bounds = np.concatenate([[0], cuts + 1, [len(st)]])
runs = [(a, b) for a, b in zip(bounds[:-1], bounds[1:]) if b - a >= 500 and sum('k_slot' in nm for nm in names[a:b]) > 0.8 * (b - a)]
if runs:
    rb = sum((en[a:b] - st[a:b]).sum() for a, b in runs); rs = sum(en[a:b].max() - st[a] for a, b in runs)
    print('  inside the %d solves (runs of >= 500 dispatches without a gap above 200 us): span %.2f ms, busy %.2f ms (%.1f %%), %d dispatches' % (
        len(runs), rs / 1e6, rb / 1e6, 100.0 * rb / rs, sum(b - a for a, b in runs)))
dur = en - st
for key in ('k_slot_a', 'k_slot_b', 'k_slot1'):
    d = np.array([dur[i] for i, nm in enumerate(names) if key in nm])
    if len(d) == 0:
        continue
    print('  %s: %d launches, total %.2f ms, median %.2f us; under 3 us (nothing to do): %d launches, %.2f ms' % (key, len(d), d.sum() / 1e6, np.median(d) / 1e3, (d < 3000).sum(), d[d < 3000].sum() / 1e6))
    print('    duration histogram (us):', ' '.join('%g-%g:%d' % (a, b, ((d >= a * 1e3) & (d < b * 1e3)).sum()) for a, b in ((0, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10), (10, 12), (12, 1e6))))
big = np.argsort(-gaps)[:12]
print('  largest gaps (us) and the kernel that followed:', ', '.join('%.0f->%s' % (gaps[i] / 1e3, names[i + 1].split('(')[0].split('::')[-1]) for i in big))
