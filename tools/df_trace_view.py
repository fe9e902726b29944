#!/usr/bin/env python3
"""tools/df_trace_view.py [npz] -- per-column view of a raw dataflow trace (tools/df_trace.py --raw)."""
import sys
import numpy as np
d = np.load(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/df_trace_raw.npz'); tasks = d['tasks']; chain = d['chain']; T = d['plan']
t0 = min(tasks[:, 0].min(), chain[:, 0].min()); us = lambda x: (x - t0) / 100.0
start, acc, done = us(tasks[:, 0]), us(tasks[:, 1]), us(tasks[:, 2]); cin, cout = us(chain[:, 0]), us(chain[:, 1])
I, J, kc = T[:, 0], T[:, 1], T[:, 3]
last = T[:, 4] == T[:, 5] - 1
idx = {(int(I[i]), int(J[i])): i for i in range(len(T)) if last[i]}
cols = [int(a) for a in sys.argv[2:]] or [30, 58, 59, 60, 61, 100]
for j in cols:
    s = idx[(j, j - 1)]; p = idx[(j, j)]
    print('J=%d chain in %.1f out %.1f (dur %.1f) | PD: acc %.1f done %.1f | T(j,j-1): acc_done %.1f done %.1f | prev chain out %.1f -> T done +%.1f, chain out +%.1f'
          % (j, cin[j], cout[j], cout[j] - cin[j], acc[p], done[p], acc[s], done[s], cout[j - 1], done[s] - cout[j - 1], cout[j] - cout[j - 1]))
    st = (tasks[s, 4:8] - t0) / 100.0
    print('     T(j,j-1) saw panels at', ' '.join('%.1f' % v for v in st), '| steps relative to potrf(j-1) out:', ' '.join('%+.1f' % (v - cout[j - 1]) for v in st), '| done %+.1f' % (done[s] - cout[j - 1]))
    for ii in (j + 1, j + 2, j + 5):
        if (ii, j - 1) in idx:
            t = idx[(ii, j - 1)]; print('     T(%d,%d): acc_done %.1f done %.1f fin %.1f (after chain out %+.1f)' % (ii, j - 1, acc[t], done[t], done[t] - acc[t], done[t] - cout[j - 1]))
fin = (done - acc)[(I != J) & last]
print('finalize dur percentiles 10/50/90/99', np.percentile(fin, [10, 50, 90, 99]))
per = np.diff(cout)
print('chain period percentiles 10/50/90', np.percentile(per, [10, 50, 90]), 'sum', per.sum())
# how late is the sub tile relative to chain-out of previous
late = np.array([done[idx[(j, j - 1)]] - cout[j - 1] for j in range(1, len(cin)) if (j, j - 1) in idx])
print('T(j,j-1) done after potrf(j-1) out: 10/50/90', np.percentile(late, [10, 50, 90]))
gap = np.array([cout[j] - done[idx[(j, j - 1)]] for j in range(1, len(cin)) if (j, j - 1) in idx])
print('potrf(j) out after T(j,j-1) done: 10/50/90', np.percentile(gap, [10, 50, 90]))
accl = np.array([acc[idx[(j, j - 1)]] - cout[j - 1] for j in range(1, len(cin)) if (j, j - 1) in idx])
print('T(j,j-1) contraction done relative to potrf(j-1) out: 10/50/90', np.percentile(accl, [10, 50, 90]))
