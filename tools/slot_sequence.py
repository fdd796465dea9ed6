"""Durations of consecutive k_slot1 launches of one warm solve (rocprofv3 kernel trace): what does each launch of an ADMM iteration cost
in sequence?   python tools/slot_sequence.py <kernel_trace.csv> [count]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 90
sl = [(a, b) for a, b, n in rows if 'k_slot1' in n]
# the LAST solve: take the final 40 % of the slot launches, print `cnt` durations from its middle
mid = int(len(sl) * 0.8)
seq = sl[mid:mid + cnt]
print(' '.join('%.1f' % ((b - a) / 1e3) for a, b in seq))
import collections
h = collections.Counter(int((b - a) / 1e3) for a, b in sl[int(len(sl) * 0.6):])
print('histogram (us: count) over the last 40 %% of the launches:', sorted(h.items()))
print('total k_slot1 time of those launches: %.2f ms over %d launches' % (sum(b - a for a, b in sl[int(len(sl) * 0.6):]) / 1e6, len(sl) - int(len(sl) * 0.6)))
