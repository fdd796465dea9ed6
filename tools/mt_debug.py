"""Which step of tests/test_gpu_multithread.py's API scenario differs under concurrency, and by how much.
    python tools/mt_debug.py [rounds] [threads]"""
import os, sys, warnings
from multiprocessing.pool import ThreadPool
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'osqp-python_amd'), os.path.join(ROOT, 'tests'), ROOT]
warnings.simplefilter('ignore')
import numpy as np
import test_gpu_multithread as T

NAMES = ['solve', 'update q', 'update bounds', 'warm start', 'update matrices+polish', 'small solve+polish', 'batch 1', 'batch 2']
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seeds = list(range(70, 76))
serial = [T._api_scenario(s) for s in seeds]
serial2 = [T._api_scenario(s) for s in seeds]
bad = 0
for a, b, s in zip(serial, serial2, seeds):
    for k, ((ia, xa, ya), (ib, xb, yb)) in enumerate(zip(a, b)):
        if ia != ib or not np.array_equal(xa, xb) or not np.array_equal(ya, yb):
            print('SERIAL repeat differs: seed %d step %d (%s): iter %d vs %d, |dx| %.3e |dy| %.3e' % (s, k, NAMES[k], ia, ib, np.abs(xa - xb).max(), np.abs(ya - yb).max()), flush=True)
for rnd in range(rounds):
    with ThreadPool(threads) as pool:
        thr = pool.map(T._api_scenario, seeds)
    for a, b, s in zip(serial, thr, seeds):
        for k, ((ia, xa, ya), (ib, xb, yb)) in enumerate(zip(a, b)):
            if ia != ib or not np.array_equal(xa, xb) or not np.array_equal(ya, yb):
                bad += 1
                print('round %d seed %d step %d (%s): iter %d vs %d, |dx| %.3e |dy| %.3e' % (rnd, s, k, NAMES[k], ia, ib, np.abs(xa - xb).max(), np.abs(ya - yb).max()), flush=True)
print('rounds %d threads %d: %d differing steps' % (rounds, threads, bad))
