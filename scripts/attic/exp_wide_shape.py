import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from qinco_amd import QincoConfig, QincoEngine, synth_state_dict, synth_vectors
for kw, diag in ((dict(D=128, de=256, dh=512, L=8), None), (dict(D=128, de=256, dh=512, L=8), "tile16")):
    cfg = QincoConfig(M=4, K=256, A=16, B=8, **kw)
    sd = synth_state_dict(cfg, 3)
    if diag:
        from qinco_amd import _lib
        from qinco_amd.build import INST, instance_cmd, hipcc, _run
        so = INST / "inst_128_256_512_48_196.so"
        cmd = [c for c in instance_cmd(hipcc(), (128, 256, 512, 48, 196), so, extra=("-DQINCO_INSTANCE_MODULE", "-shared")) if c != "-c"]
        _run(cmd)
        _lib.check(_lib.load().qinco_load_instance(str(so).encode()))
    eng = QincoEngine(cfg, sd, max_batch=16384, diagnostics={"mlp_variant": (48, 196)} if diag else None)
    x = torch.from_numpy(synth_vectors(cfg, sd, 16384, seed=1)).cuda()
    eng.encode(x); torch.cuda.synchronize()
    eng.profile_enable(True); eng.profile_read()
    t0 = time.perf_counter()
    for _ in range(2): c = eng.encode(x)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pr = eng.profile_read()
    print(eng.describe(), f"{2*16384/dt:.0f} vec/s  mlp {pr['mlp_flops']/pr['mlp_ms']/1e9:.1f} TFLOP/s algorithmic")
    eng.close()
