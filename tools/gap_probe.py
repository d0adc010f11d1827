"""Idle time on the GPU timeline of the train step, from a rocprofv3 kernel trace (csv).
usage: python tools/gap_probe.py <..._kernel_trace.csv> [min_gap_us]
Prints the total busy / idle time of the last steps and the largest gaps with the kernels on either side."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# steps are delimited by the eigensolver (one launch per step)
marks = [i for i, e in enumerate(ev) if "sym_eig" in e[2]]
if len(marks) < 3:
    sys.exit("need at least three steps in the trace")
lo, hi = marks[-3], marks[-1]
seg = ev[lo:hi]
span = seg[-1][1] - seg[0][0]
busy, gaps, end = 0, [], seg[0][0]
for s, e, n in seg:
    if s > end:
        gaps.append((s - end, prev, n))
    busy += max(0, e - max(s, end))
    end = max(end, e)
    prev = n
nsteps = 2
print(f"{nsteps} steps: span {span / 1e6 / nsteps:.3f} ms/step, busy {busy / 1e6 / nsteps:.3f}, idle {(span - busy) / 1e6 / nsteps:.3f}, "
      f"{len(seg) // nsteps} launches/step")
hist = {}
for g, a, b in gaps:
    k = "<3us" if g < 3000 else "<6us" if g < 6000 else "<15us" if g < 15000 else "<50us" if g < 50000 else ">=50us"
    hist.setdefault(k, [0, 0]); hist[k][0] += 1; hist[k][1] += g
for k, (c, t) in hist.items():
    print(f"  gaps {k:>6}: {c // nsteps:5d} per step, {t / 1e6 / nsteps:.3f} ms/step")
for g, a, b in sorted(gaps, reverse=True)[:25]:
    if g >= thr * 1000:
        print(f"  {g / 1e3:8.1f} us  after {a[:60]:60s} before {b[:60]}")
