"""Per-kernel time of ONE optimisation phase from a rocprofv3 --kernel-trace database (rocpd sqlite) of bench.py.

    rocprofv3 --kernel-trace -d out -o b -- python bench.py --steps 3 --warmup 3 --no-rooflines
    python tools/train_phase_breakdown.py out/b_results.db [csv_out]

The optimisation phases are the gaps between rollouts (a rollout = a run of rollout step kernels); the last complete gap is
reported: kernels grouped by name, sorted by total time, with the share of the phase's wall time the device was busy.
"""
import csv, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "rollout_trxl_kernel" in r[0] or "rollout_group_kernel" in r[0] or "rollout_policy_kernel" in r[0]]
gaps = [(rows[b][1] - rows[a][2], a, b) for a, b in zip(marks, marks[1:]) if rows[b][1] - rows[a][2] > 30e6]
if not gaps:
    sys.exit("no optimisation phase found between rollouts")
_, a, b = gaps[-1]
win = rows[a + 1:b]
wall = rows[b][1] - rows[a][2]
agg = {}
busy = 0
for name, s, e in win:
    k = agg.setdefault(name, [0, 0])
    k[0] += 1
    k[1] += e - s
    busy += e - s
print(f"optimisation phase: wall {wall / 1e6:.1f} ms, kernels {len(win)}, summed kernel time {busy / 1e6:.1f} ms")
out = sorted(agg.items(), key=lambda kv: -kv[1][1])
for name, (n, t) in out[:45]:
    print(f"{t / 1e6:8.2f} ms {100 * t / busy:5.1f}%  {n:6d} x {t / n / 1e3:8.1f} us  {name[:120]}")
if len(sys.argv) > 2:
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for name, (n, t) in out:
            w.writerow([name, n, t, t / n, 100 * t / busy])

# the kernel sequence of ONE minibatch step (between two optimiser launches), for reading the launch structure
opt = [i for i, (name, s, e) in enumerate(win) if "adamw_clip_kernel" in name]
if len(opt) > 3 and "--sequence" in sys.argv:
    seq = win[opt[-3] + 1:opt[-2] + 1]
    t0 = seq[0][1]
    print(f"\none minibatch step: {len(seq)} kernels, {(seq[-1][2] - t0) / 1e3:.1f} us")
    for name, s, e in seq:
        print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f}  {name[:100]}")
