"""Kernel timeline of ONE rollout step from a rocprofv3 kernel trace of tools/rollout_profile.py (evidence for profiles/).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tp -- python tools/rollout_profile.py
    python tools/rollout_timeline.py /tmp/prof/tp_kernel_trace.csv > profiles/r01_rollout_step_timeline.txt
"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "rollout_window_kernel" in r["Kernel_Name"]]
a, b = idx[700], idx[701]          # a step in the middle of the second rollout
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
def short(n):
    if "anonymous namespace" in n and "at::native" not in n:
        return n.split("::")[1].split("(")[0][:40]
    if n.startswith("Cijk"):
        return "library GEMM " + n.split("_MT")[1][:14]
    if "at::native" in n:
        return "aten " + n.split("at::native::")[1][:44]
    return n[:50]
print("one rollout step (rocprofv3 serialises kernels; durations include the per-kernel dispatch overhead)")
print("  start_us  dur_us  kernel")
for r in seg:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {short(r['Kernel_Name'])}  [grid {r['Grid_Size_X']} x wg {r['Workgroup_Size_X']}]")
print(f"period to the next step's first kernel: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us; kernels in the step: {len(seg)}")
