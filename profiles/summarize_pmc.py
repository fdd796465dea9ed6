#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: mean counter value per dispatch, split into 'active'
dispatches and early-exit (no-op) ones by a duration-free criterion (counter value above 1% of the kernel's max)."""
import csv
import hashlib
import os
import re
import collections
import sys


def source_sha16(workload):
    """what the counters describe: the kernel sources of this workload as they are in this tree (bench.py pmc_sources / pmc_traffic: a summary taken of other
    code is refused)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = ['pcg_hip.hip', 'hip_common.h']
    if workload.startswith('lasso'):
        files.append('woodbury_hip.hip')
    if workload.startswith('portfolio'):
        files.append('wbdirect_hip.hip')
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(root, 'osqp-python_amd', 'csrc', f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


path, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
with open(path) as f:
    rd = csv.DictReader(f)
    kcol = [c for c in rd.fieldnames if 'kernel' in c.lower() and 'name' in c.lower()]
    print('columns:', rd.fieldnames, '-> kernel column', kcol)
    for row in rd:
        name = row[kcol[0]] if kcol else ''
        mm = re.search(r'(k_\w+(<\d+>)?)', name)
        name = mm.group(1) if mm else name.split('(')[0]
        agg[name][row['Counter_Name']].append(float(row['Counter_Value']))
with open(out, 'w') as g:
    g.write('kernel,counter,dispatches,mean_all,active_dispatches,mean_active,max\n')
    for k in sorted(agg):
        for c, v in agg[k].items():
            mx = max(v)
            act = [x for x in v if x > 0.01 * mx] if mx > 0 else []
            g.write('%s,%s,%d,%.6g,%d,%.6g,%.6g\n' % (k, c, len(v), sum(v) / len(v), len(act), (sum(act) / len(act)) if act else 0.0, mx))
    wl = re.search(r'_pmc_(.+)_(FETCH_SIZE|WRITE_SIZE)\.csv$', os.path.basename(out))
    g.write('__source__,%s,0,0,0,0,0\n' % source_sha16(wl.group(1) if wl else ''))      # (last row: which kernel sources were measured)
print(open(out).read())
