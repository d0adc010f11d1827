"""Reads a rocprofv3 kernel trace (CSV) of bench.py and, for every train step (from one mbvae noise kernel to the next),
prints the wall time, the time with no kernel on the GPU, the time with kernels of two queues at once, and the busy time
per queue; then the largest idle gaps of a chosen step with the kernels around them.

    python tools/timeline_gaps.py <kernel_trace.csv> [step index to detail]
"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "noise_nodes_kernel" in r[3]]
detail = int(sys.argv[2]) if len(sys.argv) > 2 else -1
for si in range(len(marks) - 1):
    seg = rows[marks[si]:marks[si + 1]]
    t0, t1 = seg[0][0], rows[marks[si + 1]][0]
    ev = []
    for s, e, q, _ in seg:
        ev.append((s, 1, q)); ev.append((min(e, t1), -1, q))
    ev.sort()
    active = defaultdict(int)
    last = t0
    idle = both = 0
    busy_q = defaultdict(int)
    for t, d, q in ev:
        n_q = sum(1 for v in active.values() if v > 0)
        if n_q == 0:
            idle += t - last
        if n_q >= 2:
            both += t - last
        for qq, v in active.items():
            if v > 0:
                busy_q[qq] += t - last
        active[q] += d
        last = t
    qs = ", ".join(f"q{q}: {v / 1e6:.2f}" for q, v in sorted(busy_q.items()))
    print(f"step {si:2d}: {len(seg):4d} kernels, wall {(t1 - t0) / 1e6:7.2f} ms, idle {idle / 1e6:6.2f} ms, two queues busy {both / 1e6:6.2f} ms, busy per queue [{qs}]")
    if si == detail:
        # gaps: intervals with nothing running
        cur_end = seg[0][1]
        prev = seg[0]
        gaps = []
        for s, e, q, name in seg[1:]:
            if s > cur_end:
                gaps.append((s - cur_end, prev[3][:60], name[:60]))
            if e > cur_end:
                cur_end = e; prev = (s, e, q, name)
        gaps.sort(reverse=True)
        print(f"  {len(gaps)} idle gaps, sum {sum(g[0] for g in gaps) / 1e6:.2f} ms; largest:")
        for gsz, a, b in gaps[:25]:
            print(f"   {gsz / 1e3:8.1f} us  after {a}  before {b}")
