import sys, torch
sys.path.insert(0, "/root/repo")
from open_provence_amd.engine import HipEncoder
from open_provence_amd.synthetic import named_dims, refinit_state_dict
for model in ("xsmall", "base", "en-gte", "large"):
    for w in ("fp32", "bf16"):
        dims = named_dims(model)
        state = refinit_state_dict(dims, seed=7)
        if w == "bf16":
            state = {k: (v.to(torch.bfloat16).to(torch.float32) if v.ndim == 2 and "embeddings" not in k else v) for k, v in state.items()}
        enc = HipEncoder(dims, device="cuda:0")
        enc.load_state_dict(state)
        c = enc.calibration
        print(model, w, c["chosen_set"], "default", c["default_set"], {k: f"{v:.2e}" for k, v in c["candidates"].items()}, flush=True)
        enc.close()
