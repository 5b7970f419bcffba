"""dev: HIP vs oracle gradients of the tests/test_dist_gpu.py model for individual batch seeds."""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
    sys.path.insert(0, p)
import vct_oracle as O
from helpers import build_model
from test_dist_gpu import MC, VOCAB, _batch
cfg = O.cfg_from_model_config(MC, VOCAB)
torch.manual_seed(50)
m = build_model(MC, VOCAB, "cuda", torch.float32)
m.train()
sd = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
for seed in range(10, 18):
    f, mk, ids = _batch(seed, "cuda")
    m._ps.refresh_shadow(force=True)
    m.train_step_kernels(f, mk, ids)
    torch.cuda.synchronize()
    ref = O.caption_loss_and_grads(sd, cfg, f.cpu().numpy(), mk.cpu().numpy(), ids.cpu().numpy())[1]
    worst = max(((float(np.linalg.norm(m._ps.g[k].double().cpu().numpy() - g) / max(np.linalg.norm(g), 1e-30)), k) for k, g in ref.items()))
    print(seed, worst, ids[:, :].tolist()[2])
