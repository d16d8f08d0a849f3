"""Driver for rocprofv3 traces of small calls: python scripts/prof_calls.py <workload> <mode> <rows> [reps]
mode: decode | encode | encode1 (greedy, B = 1).  Distinct resident inputs per call; prints vectors/s."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    from qinco_amd import QincoEngine, synth_state_dict, synth_vectors
    from qinco_amd.config import BASELINE_CONFIGS
    wl, mode, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    cfg = BASELINE_CONFIGS[wl]
    sd = synth_state_dict(cfg, 1236)
    eng = QincoEngine(cfg, sd, max_batch=max(n, 1024))
    if mode == "encode1":
        eng.set_beam(cfg.A, 1)
    x = torch.from_numpy(synth_vectors(cfg, sd, n * 4, seed=7)).cuda()
    rs = np.random.RandomState(3)
    codes = torch.from_numpy(np.stack([rs.randint(0, k, size=n * 4) for k in cfg.K_vals], axis=1).astype(np.int32)).cuda()

    def call(i):
        o = (i % 4) * n
        if mode == "decode":
            eng.decode(codes[o:o + n], check=False)
        else:
            eng.encode(x[o:o + n], code_dtype=np.uint8 if not cfg.ivf else np.int32)
    call(0)
    torch.cuda.synchronize()
    eng.profile_enable(True)
    eng.profile_read()
    t0 = time.perf_counter()
    for i in range(reps):
        call(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pr = eng.profile_read()
    print(f"{wl} {mode} rows={n}: {reps * n / dt:.0f} vectors/s, {dt / reps * 1e6:.1f} us per call")
    # what the library says the matrix pipe executed (qinco_profile_read2) -- the figure the SQ_INSTS_VALU_MFMA_MOPS_F32 pass checks
    import json
    print("LIBRARY " + json.dumps({"workload": wl, "mode": mode, "rows": n, "calls_counted": reps, "mlp_launches": pr["mlp_launches"],
                                  "executed_flops_per_call": pr["mlp_flops_executed"] / reps, "algorithmic_flops_per_call": pr["mlp_flops"] / reps,
                                  "mlp_ms_per_call": pr["mlp_ms"] / reps}))


if __name__ == "__main__":
    main()
