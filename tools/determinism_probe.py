#!/usr/bin/env python
"""Two fresh models, same seed, N recorded training steps at the bench batch: where do they first differ?  Dev tool."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from helpers import build_model, load_golden, model_config_of  # noqa: E402
import vct_oracle as O  # noqa: E402
from vct_amd.trainer import CaptionTrainer, FusedAdam  # noqa: E402

DEV = torch.device("cuda", 0)
from vct_amd import engine as _eng
if os.environ.get('OVERLAP_DW', '1') == '0': _eng._StackBase.overlap_dw = False
if os.environ.get('GROUP_DW', '1') == '0': _eng._StackBase.group_dw = False
if os.environ.get('OVERLAP_KV', '1') == '0': _eng._StackBase.overlap_kv = False
mc = model_config_of(load_golden("cfgA_slices.npz"))
V = 30522
cfg = O.cfg_from_model_config(mc, V)
p = O.init_params(cfg, seed=31)
f, mk, ids = O.synthetic_batch(256, 12, 512, 20, V, seed=5)
feats, mask, idt = torch.from_numpy(f).to(DEV), torch.from_numpy(mk).to(DEV), torch.from_numpy(ids).to(DEV)
nsteps = int(os.environ.get("STEPS", "3"))
use_list = os.environ.get("LIST", "1") == "1"
for rep in range(int(os.environ.get("REPS", "4"))):
    outs = []
    for _ in range(2):
        mm = build_model(dict(mc, dropout=float(os.environ.get("DROP", "0.3"))), V, DEV, torch.bfloat16, p)
        mm.train(); mm._seed.fill_(99)
        if os.environ.get('OVERLAP_ENC', '1') == '0': mm.overlap_enc_bwd = False
        tr = CaptionTrainer(mm, FusedAdam(mm, lr=1e-4), launch_list=use_list)
        tr.adam_after_backward = os.environ.get('AAB', '0') == '1'
        losses, grads, pars, shads = [], [], [], []
        for s in range(nsteps):
            losses.append(tr.step(feats, mask, idt).clone())
            grads.append(mm.flat_grads.clone()); pars.append(mm.flat_params.clone()); shads.append(mm._ps.cflat.clone())
            wt = mm._ps.transposed.get('cap_decoder.generator.weight'); shads.append(wt[0].clone() if wt else torch.zeros(1))
        torch.cuda.synchronize()
        outs.append((torch.cat(losses), grads, mm.flat_params.clone(), mm._ps, pars, shads))
    (la, ga, pa, ps, para, sha), (lb, gb, pb, _, parb, shb) = outs
    msg = [f"rep {rep}: losses equal {bool(torch.equal(la, lb))} params equal {bool(torch.equal(pa, pb))}"]
    for s in range(nsteps):
        if not torch.equal(ga[s], gb[s]):
            bad = (ga[s] != gb[s]).nonzero().flatten()
            names = []
            for n in ps.names:
                o, k = ps.offsets[n], ps.params[n].numel()
                c = int(((bad >= o) & (bad < o + k)).sum())
                if c:
                    names.append(f"{n}:{c}")
            msg.append(f"  step {s}: {bad.numel()} gradient elements differ: {names[:8]}")
            break
    for s_ in range(nsteps):
        if not torch.equal(para[s_], parb[s_]):
            bad = (para[s_] != parb[s_]).nonzero().flatten()
            names = []
            for n in ps.names:
                o, k = ps.offsets[n], ps.params[n].numel()
                c = int(((bad >= o) & (bad < o + k)).sum())
                if c:
                    names.append(f"{n}:{c}")
            msg.append(f"  params after step {s_}: {bad.numel()} differ: {names[:6]}")
            o = ps.offsets["cap_decoder.generator.weight"]
            rc = [((int(i) - o) // 512, (int(i) - o) % 512) for i in bad[:40].tolist()]
            msg.append(f"  (row, col) in generator.weight: {rc}")
            ia = int(bad[0]); msg.append(f"  values: {float(para[s_][ia])!r} vs {float(parb[s_][ia])!r}")
            break
    for s_ in range(nsteps):
        e1, e2 = torch.equal(sha[2 * s_], shb[2 * s_]), torch.equal(sha[2 * s_ + 1], shb[2 * s_ + 1])
        if not (e1 and e2):
            msg.append(f"  after step {s_}: bf16 shadow equal {e1}, W_g^T equal {e2} ({int((sha[2*s_+1] != shb[2*s_+1]).sum())} elements)")
            break
    print("\n".join(msg), flush=True)
