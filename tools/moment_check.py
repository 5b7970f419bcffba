#!/usr/bin/env python
"""After the FIRST overlapped training step from zero moments, Adam's first / second moments must equal (1 - b1) g and (1 - b2) g^2 of the
step's own gradient, element for element: any other value means somebody else wrote the buffers (or Adam read a gradient that was
still being written).  Dev tool."""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "oracle")]
from helpers import build_model, load_golden, model_config_of  # noqa: E402
import vct_oracle as O  # noqa: E402
from vct_amd.trainer import CaptionTrainer, FusedAdam  # noqa: E402

DEV = torch.device("cuda", 0)
mc = model_config_of(load_golden("cfgA_slices.npz"))
V = 30522
cfg = O.cfg_from_model_config(mc, V)
p = O.init_params(cfg, seed=31)
f, mk, ids = O.synthetic_batch(256, 12, 512, 20, V, seed=5)
feats, mask, idt = torch.from_numpy(f).to(DEV), torch.from_numpy(mk).to(DEV), torch.from_numpy(ids).to(DEV)
c1 = torch.tensor(1.0, dtype=torch.float32) - torch.tensor(0.9, dtype=torch.float32)
c2 = torch.tensor(1.0, dtype=torch.float32) - torch.tensor(0.999, dtype=torch.float32)
for rep in range(int(os.environ.get("REPS", "6"))):
    mm = build_model(dict(mc, dropout=float(os.environ.get("DROP", "0.3"))), V, DEV, torch.bfloat16, p)
    mm.train(); mm._seed.fill_(99)
    opt = FusedAdam(mm, lr=1e-4)
    tr = CaptionTrainer(mm, opt, launch_list=os.environ.get("LIST", "1") == "1")
    tr.step(feats, mask, idt)
    torch.cuda.synchronize()
    e = mm.caption_param_end
    g = mm.flat_grads[:e]
    m_ref = g * c1.to(DEV)
    v_ref = (g * c2.to(DEV)) * g
    bm = (opt.exp_avg[:e] != m_ref).nonzero().flatten()
    bv = (opt.exp_avg_sq[:e] != v_ref).nonzero().flatten()
    def where(bad):
        out = []
        ps = mm._ps
        for n in ps.names:
            o, k = ps.offsets[n], ps.params[n].numel()
            c = int(((bad >= o) & (bad < o + k)).sum())
            if c:
                out.append(f"{n}:{c}")
        return out[:6]
    print(f"rep {rep}: first-moment mismatches {bm.numel()} {where(bm)}; second-moment mismatches {bv.numel()} {where(bv)}", flush=True)
