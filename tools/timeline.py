#!/usr/bin/env python3
"""Overlap of the streams' kernels in a rocprofv3 kernel trace:  tools/timeline.py <kernel_trace.csv> [first_fraction last_fraction]
(the CSV of `rocprofv3 --kernel-trace -f csv`).  Prints, for the middle of the run (default: 20 % .. 95 % of the dispatches by
start time, i.e. the timed steps without warm-up and without the one-stream passes behind them): the wall span, the time in which
NO kernel runs, the time-weighted number of kernels in flight, and per kernel the sum of its durations as a share of the span (the
sum over kernels exceeds 1 when kernels overlap)."""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
ev = ev[int(len(ev) * lo):int(len(ev) * hi)]
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = []
for s, e, _ in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
idle = 0; weighted = 0; level = 0; last = t0; hist = collections.Counter()
for t, d in pts:
    dt = t - last
    if level == 0: idle += dt
    weighted += level * dt; hist[min(level, 6)] += dt
    level += d; last = t
span = t1 - t0
print(f"dispatches {len(ev)}, span {span / 1e3:.1f} us, idle (no kernel running) {idle / 1e3:.1f} us = {idle / span:.3f}, kernels in flight (time-weighted) {weighted / span:.2f}")
print("time share by kernels in flight: " + ", ".join(f"{k}{'+' if k == 6 else ''}: {v / span:.3f}" for k, v in sorted(hist.items())))
per = collections.Counter(); cnt = collections.Counter()
for s, e, k in ev:
    name = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    per[name] += e - s; cnt[name] += 1
for k, v in per.most_common(16):
    print(f"  {k:44s} {cnt[k]:5d} x {v / cnt[k] / 1e3:8.1f} us  = {v / span:.3f} of the span")
print(f"  sum of kernel time / span = {sum(per.values()) / span:.2f}")

# time in which only the small latency-bound kernels run (the chip is mostly empty then) and which kernels run ALONE
SMALL = ("select_kernel", "emit_kernel", "angle_tail_kernel")
pts2 = []
for s, e, k in ev:
    name = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    pts2.append((s, 1, name)); pts2.append((e, -1, name))
pts2.sort(key=lambda t: (t[0], t[1]))
running = collections.Counter(); last = t0; small_only = 0; alone = collections.Counter(); pairs = collections.Counter()
for t, d, name in pts2:
    dt = t - last
    act = [k for k, v in running.items() if v > 0]
    if act and all(a in SMALL for a in act): small_only += dt
    if len(act) == 1 and sum(running.values()) == 1: alone[act[0]] += dt
    if sum(running.values()) == 2: pairs[" + ".join(sorted(k for k, v in running.items() for _ in range(v)))] += dt
    running[name] += d; last = t
print(f"only {', '.join(SMALL)} running: {small_only / 1e3:.1f} us = {small_only / span:.3f} of the span")
print("running alone: " + ", ".join(f"{k} {v / span:.3f}" for k, v in alone.most_common(8)))
print("pairs: " + ", ".join(f"{k} {v / span:.3f}" for k, v in pairs.most_common(8)))
