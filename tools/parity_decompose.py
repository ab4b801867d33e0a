"""Build-container side of the round-4 parity decomposition (scratch data; nothing here is shipped in the product).

    ship <case>      tools/scratch/parity/ref_full_<case>_u0.npz (written by `ETM_GOLDEN_FULL_DUMP=tools/scratch/parity
                     python tests/golden/make_golden.py rollout:<case>`) -> tools/scratch/parity/ship/xgrad_<case>_u0s0.npz:
                     the reference's fp32 gradient and the float64 evaluation of step 0 as whole tensors, for tools/parity_probe.py
    moves <case> <variant>   element-wise comparison of the HIP path's parameters after every optimiser step of the first update
                     (gpurun_out/parity/hip_params_*.npz) with the reference's: which elements carry the movement error, and what their
                     gradients look like
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCR = os.path.join(REPO, "tools", "scratch", "parity")


def ship(name):
    z = np.load(os.path.join(SCR, f"ref_full_{name}_u0.npz"))
    out = {}
    for k in z.files:
        if k.startswith("s0/xgrad/"):
            out["xgrad/" + k[len("s0/xgrad/"):]] = z[k]
        elif k.startswith("s0/grad/"):
            out["grad/" + k[len("s0/grad/"):]] = z[k]
    os.makedirs(os.path.join(SCR, "ship"), exist_ok=True)
    path = os.path.join(SCR, "ship", f"xgrad_{name}_u0s0.npz")
    np.savez(path, **out)
    print(path, os.path.getsize(path) / 2 ** 20, "MiB")


def moves(name, variant):
    ref = np.load(os.path.join(SCR, f"ref_full_{name}_u0.npz"))
    keys = [k[len("s0/grad/"):] for k in ref.files if k.startswith("s0/grad/")]
    s = 0
    while os.path.exists(os.path.join(REPO, "gpurun_out", "parity", f"hip_params_{name}_{variant}_s{s}.npz")):
        hip = np.load(os.path.join(REPO, "gpurun_out", "parity", f"hip_params_{name}_{variant}_s{s}.npz"))
        num = den = 0.0
        rows = []
        for k in keys:
            p0 = ref[f"s0/params/{k}"].astype(np.float64)
            r = ref[f"s{s}/sd_after/{k}"].astype(np.float64)
            h = hip[k].astype(np.float64)
            e, m = float(np.sum((h - r) ** 2)), float(np.sum((r - p0) ** 2))
            rows.append((k, (e / max(m, 1e-300)) ** 0.5, e, m))
            num, den = num + e, den + m
        print(f"after step {s}: movement error (whole tensors, from the initial parameters) {np.sqrt(num / den):.3e}")
        for k, rel, e, m in sorted(rows, key=lambda t: -t[2])[:8]:
            print(f"    {k:58s} rel {rel:.2e}  share of the squared error {e / num:.2f}")
        s += 1


if __name__ == "__main__":
    globals()[sys.argv[1]](*sys.argv[2:])
