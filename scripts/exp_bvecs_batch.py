"""The .bvecs -> encode_database -> part file path of bench.py's encode_db_bvecs_qinco2S leg against the host batch size."""
import sys, json
sys.path.insert(0, "/root/repo")
import torch
import bench
dev = torch.device("cuda:0")
for mult in (4, 8, 16):
    import qinco_amd.encode_db as E
    orig = E.encode_database
    def patched(model, db, out, **kw):
        kw["batch"] = mult * 16384
        return orig(model, db, out, **kw)
    E.encode_database = patched
    r = bench.leg_encode_db_bvecs(torch, dev, 1_000_000, 16384, "S")
    E.encode_database = orig
    print(json.dumps({"host_batch": mult * 16384, "value": round(r["value"]), "host_over_resident": round(r["host_over_resident"], 4), "seconds": round(r["seconds"], 3)}), flush=True)
