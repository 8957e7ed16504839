import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from open_provence_amd import _lib
from open_provence_amd.engine import HipEncoder
from open_provence_amd.packing import pack_rows
from open_provence_amd.synthetic import named_dims, refinit_state_dict, synth_pair_batch, synth_state_dict

for n_layers in (2, 3, 10):
    dims = named_dims("xsmall", num_layers=n_layers)
    for init in ("refinit", "o1"):
        state = refinit_state_dict(dims, seed=1234) if init == "refinit" else synth_state_dict(dims, seed=1234)
        rows = synth_pair_batch(dims, 288, 512, seed=1234)
        ids_np, cu_np, max_len = pack_rows(rows)
        ids, cu = torch.from_numpy(ids_np).cuda(), torch.from_numpy(cu_np).cuda()
        outs = {}
        for label, flags in (("pairs", 0), ("pairs2", 0), ("8x16", _lib.OP_FLAG_NO_LAYER_PAIRS), ("x3", _lib.OP_FLAG_NO_LAYER_PAIRS)):
            enc = HipEncoder(dims, device="cuda:0", flags=flags)
            enc.load_state_dict(state, calibrate=False, kernel_set="bf16x3" if label == "x3" else "f16")
            p, r = enc.forward_packed(ids, cu, cu_np, max_len)
            torch.cuda.synchronize()
            outs[label] = (p.cpu().numpy(), r.cpu().numpy())
            enc.close()
        d = lambda a, b: max(np.abs(outs[a][0] - outs[b][0]).max(), np.abs(outs[a][1] - outs[b][1]).max())
        print(f"layers {n_layers:2d} {init:8s} pairs-vs-8x16 {d('pairs','8x16'):.3e}  pairs-vs-pairs {d('pairs','pairs2'):.3e}  pairs-vs-x3 {d('pairs','x3'):.3e}  8x16-vs-x3 {d('8x16','x3'):.3e}  |out| {np.abs(outs['x3'][0]).max():.2f}", flush=True)
