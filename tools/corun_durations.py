"""Per-kernel average duration in the two-stream steps of a bench.py kernel trace (rocprofv3 CSV), by queue: shows what the
node / global kernels cost while they share the GPU with the edge-tensor kernels (compare with bench.py's one-stream table).

    python tools/corun_durations.py <kernel_trace.csv> [first step] [last step]
"""
import csv, sys
from collections import defaultdict
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "noise_nodes_kernel" in r[3]]
a, b = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, 5)
seg = rows[marks[a]:marks[b + 1]]
nsteps = b - a + 1
qs = sorted({q for _, _, q, _ in seg})
tot = defaultdict(lambda: [0, 0])
for s, e, q, n in seg:
    t = tot[(q, n.split("(")[0].replace("void ", "")[:70])]
    t[0] += e - s; t[1] += 1
for q in qs:
    items = sorted(((v[0], k[1], v[1]) for k, v in tot.items() if k[0] == q), reverse=True)
    print(f"queue {q}: {sum(i[0] for i in items) / nsteps / 1e6:.2f} ms of kernel durations per step")
    for d, n, c in items[:14]:
        print(f"   {d / nsteps / 1e6:7.3f} ms/step  {c // nsteps:4d} calls/step  avg {d / c / 1e3:7.1f} us  {n}")
print(f"wall per step {(rows[marks[b + 1]][0] - seg[0][0]) / nsteps / 1e6:.2f} ms")
