#!/usr/bin/env python3
"""Per-kernel duration statistics from a rocprofv3 --kernel-trace CSV, separating ACTIVE dispatches from the early-exit
ones (a converged PCG turns the rest of an ADMM iteration's K1/K2/Kv launches into immediate returns; rocprofv3 --stats
averages over both).  A dispatch counts as active if it lasts longer than (min + upper-half median) / 2."""
import csv
import collections
import re
import statistics
import sys

path, out = sys.argv[1], sys.argv[2]
dur = collections.defaultdict(list)
with open(path) as f:
    rd = csv.DictReader(f)
    kcol = [c for c in rd.fieldnames if 'kernel' in c.lower() and 'name' in c.lower()][0]
    for row in rd:
        mm = re.search(r'(k_\w+(<\d+>)?)', row[kcol])
        if mm:
            dur[mm.group(1)].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
with open(out, 'w') as g:
    g.write('kernel,dispatches,mean_ns,active_dispatches,active_mean_ns,active_median_ns,noop_mean_ns\n')
    for k in sorted(dur):
        v = sorted(dur[k]); up = v[len(v) // 2:]
        thr = (v[0] + statistics.median(up)) / 2
        act = [x for x in v if x > thr]; noop = [x for x in v if x <= thr]
        g.write('%s,%d,%.0f,%d,%.0f,%.0f,%.0f\n' % (k, len(v), statistics.mean(v), len(act), statistics.mean(act) if act else 0,
                                                   statistics.median(act) if act else 0, statistics.mean(noop) if noop else 0))
print(open(out).read())
