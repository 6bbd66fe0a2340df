"""Where is the GPU idle inside a step?  Reads a rocprofv3 --kernel-trace CSV (one row per dispatch with start / end
timestamps) of `bench.py --steps S --warmup W --no-kernel-timer --no-cpu-baseline` and reports, for the last steps: wall time,
the union of the kernels' busy intervals (time with at least one kernel running), the idle remainder, the time with exactly one
/ two / three or more kernels in flight, and the longest idle gaps with the kernels on either side.
    python tools/timeline_gaps.py <kernel_trace.csv> [n_last_adam_steps]"""
import csv
import sys


def main():
    path = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # a step ends with the last adam_kernel launch of its three groups
    adam_ends = [e for s, e, n in rows if "adam_kernel" in n]
    ends = adam_ends[2::3]
    if len(ends) < nsteps + 1:
        raise SystemExit(f"only {len(ends)} steps in the trace")
    t0, t1 = ends[-nsteps - 1], ends[-1]
    ev = []
    sel = [(s, e, n) for s, e, n in rows if e > t0 and s < t1]
    for s, e, n in sel:
        ev.append((max(s, t0), 1))
        ev.append((min(e, t1), -1))
    ev.sort()
    depth, last = 0, t0
    hist = {}
    gaps = []
    for t, d in ev:
        hist[depth] = hist.get(depth, 0) + (t - last)
        if depth == 0 and t > last:
            gaps.append((t - last, last))
        depth += d
        last = t
    wall = (t1 - t0) / nsteps
    print(f"{nsteps} steps, {len(sel) / nsteps:.0f} launches per step, wall {wall / 1e6:.2f} ms per step")
    for k in sorted(hist):
        print(f"  {k} kernel(s) in flight: {hist[k] / nsteps / 1e6:8.2f} ms per step")
    ksum = sum(min(e, t1) - max(s, t0) for s, e, n in sel) / nsteps
    print(f"  sum of kernel durations {ksum / 1e6:.2f} ms per step")
    gaps.sort(reverse=True)
    print("longest idle gaps:")
    for g, at in gaps[:12]:
        before = max((r for r in sel if r[1] <= at + 1), key=lambda r: r[1], default=None)
        after = min((r for r in sel if r[0] >= at + g - 1), key=lambda r: r[0], default=None)
        print(f"  {g / 1e3:8.1f} us  after {before[2][:60] if before else '-'}  before {after[2][:60] if after else '-'}")
    small = sum(g for g, _ in gaps if g < 20000) / nsteps
    print(f"  idle in gaps < 20 us: {small / 1e6:.2f} ms per step; in gaps >= 20 us: {(hist.get(0, 0) / nsteps - small) / 1e6:.2f} ms per step")


if __name__ == "__main__":
    main()
