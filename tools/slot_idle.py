"""Where do the idle launches of a solve sit?  The k_slot1 launches of the LAST solve of a rocprofv3 kernel trace, in order, as one character per launch
(i: under 6 us = nothing to do, b: 6 - 9 us = chunk start, a: 9 - 11.6 us = KA / F_0, F: a PCG iteration) with every run of other kernels (a boundary group)
as '|', and the count of idle launches per chunk.   python tools/slot_idle.py <kernel_trace.csv>"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
# solves = runs of dispatches without a gap above 200 us that hold at least 500 k_slot1 launches
runs, cur = [], [rows[0]]
for prev, r in zip(rows, rows[1:]):
    if r[0] - prev[1] > 200000: runs.append(cur); cur = []
    cur.append(r)
runs.append(cur)
solves = [r for r in runs if sum('k_slot1' in x[2] for x in r) >= 500]
last = solves[-2] if len(solves) > 1 else solves[-1]      # (the last run of dispatches also holds the bench's timing probes)
s, chunks, idle = '', [], 0
for a, b, n in last:
    if 'k_slot1' in n:
        d = (b - a) / 1e3
        c = 'i' if d < 6 else ('b' if d < 9 else ('a' if d < 11.6 else 'F'))
        idle += c == 'i'; s += c
    elif not s.endswith('|'):
        s += '|'; chunks.append(idle); idle = 0
chunks.append(idle)
print('%d solves in the trace; the one before the last: %d k_slot1 launches, %d idle, span %.2f ms' % (len(solves), sum(ch in 'ibaF' for ch in s), s.count('i'), (last[-1][1] - last[0][0]) / 1e6))
print('idle launches per chunk (between boundary groups):', chunks)
for k in range(0, len(s), 200): print(s[k:k + 200])
