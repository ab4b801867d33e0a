"""Round 6 probe: observation rows drawn by the host STRAIGHT into device memory (large BAR) against pinned memory + copy-engine upload.

    python tools/direct_rows_probe.py [threads]

Per variant: 8 rows of 3x84x84 floats (one worker group's step) drawn by libetm_envgen.so's pool, then a kernel that reads them
(a sum), host-timed from the first draw to the kernel's completion; and the correctness of what the kernel saw, every repetition."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import numpy as np, torch
from environments import envgen
from etm import lib as etm_lib
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
lib, hip = envgen.load(required=True), etm_lib.load()
pool = lib.etm_envgen_pool_create(threads)
rows, rf = 8, 3 * 84 * 84
gens = [np.random.default_rng(w) for w in range(rows)]
d_buf = torch.zeros((rows, rf), device=dev)
pin = torch.zeros((rows, rf)).pin_memory()
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()


def run(direct, chunks, reps=200):
    states = np.ascontiguousarray(np.stack([envgen.state_of(g) for g in gens]))
    ref_states = states.copy()
    ref = np.empty((rows, rf), dtype=np.float32)
    best, tot, bad = 1e9, 0.0, 0
    per = rows // chunks
    for rep in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(chunks):
            lo = c * per
            dst = (d_buf.data_ptr() if direct else pin.data_ptr()) + lo * rf * 4
            lib.etm_pcg64_fill_rows_f32(pool, states[lo:].ctypes.data, dst, rf, per)
            if not direct:
                hip.etm_upload(d_buf.data_ptr() + lo * rf * 4, pin.data_ptr() + lo * rf * 4, per * rf * 4, st)
        s = d_buf.sum(dtype=torch.float64)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best, tot = min(best, dt), tot + dt
        lib.etm_pcg64_fill_rows_f32(None, ref_states.ctypes.data, ref.ctypes.data, rf, rows)
        if abs(float(s) - float(ref.astype(np.float64).sum())) > 1e-3 or (rep % 50 == 0 and not np.array_equal(d_buf.cpu().numpy(), ref)):
            bad += 1
    return best * 1e6, tot / reps * 1e6, bad


for direct in (False, True):
    for chunks in (1, 2):
        b, m, bad = run(direct, chunks)
        print(f"{'direct into device memory' if direct else 'pinned + etm_upload      '} {threads} threads, {chunks} chunk(s): draw -> kernel done {b:6.1f} us best, {m:6.1f} mean; wrong sums: {bad}")
