"""The rows around the hot path on the GPU: device-resident input pipeline (vct_gather_pad_rows), exact resume,
batched evaluation loop -- against the reference-recorded loader fixture and the host path."""
import json
import os

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from helpers import build_model, load_golden
from test_data_cpu import ToyPrep, ToyTok, _dataset, _same_params

pytestmark = pytest.mark.gpu
DEV = "cuda"

MC = {"modal": ["clip"], "modal_shape": [16], "text_enc_type": "CLIP", "embed_dim": 64, "dropout": 0.3, "loss_beta": 0.5,
      "matching": {"enable_tem": False, "matching_loss": "CSL"}, "activation": "gelu",
      "video_encoder": {"layer": 2, "nhead": 4, "feedforward": 128,
                        "mme": {"temporal": "encoding", "modal_different": True, "do_norm": False, "aggregation": "avg"}, "aoa": False},
      "caption_decoder": {"layer": 2, "nhead": 4, "feedforward": 128, "sce_loss_alpha": 0.5}, "pretrained_model": None}
TC = {"optimizer": {"name": "adam", "learning_rate": 1e-3, "beta": [0.9, 0.999], "weight_decay": 0,
                    "lr_scheduler": {"name": "CosineAnnealingLR", "T_max": 10, "eta_min": 1e-5}}}
VOCAB = 2000


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.fixture(scope="module")
def split(tmp_path_factory):
    z = load_golden("dataloader.npz")
    meta = json.loads(str(z["meta"]))
    d = tmp_path_factory.mktemp("split_gpu")
    os.makedirs(d / "feats")
    for v in meta["vids"]:
        np.save(d / "feats" / f"{v}.npy", z[f"clip_{v}"])
    (d / "ann.json").write_text(json.dumps(meta["annotation"]))
    (d / "msvd_train.txt").write_text(meta["msvd_train_txt"])
    return z, meta, d


def _model(dtype=torch.bfloat16, seed=11):
    torch.manual_seed(seed)
    m = build_model(MC, VOCAB, DEV, dtype)
    m.cap_preprocessor.tokenizer = ToyTok()          # strings -> ids without the HF vocabulary files
    return m


@pytest.mark.parametrize("case", ["msrvtt_train_by_caption", "msvd_train_by_caption", "msrvtt_val_by_video"])
def test_device_loader_batches_equal_reference_collate(split, case):
    """Sequential DeviceLoader batches are the reference's collate_fn batches, bit for bit (it is a copy), with the
    captions already tokenised; bf16 output is the rounded fp32 batch."""
    from vct_amd import data
    z, meta, d = split
    g = meta["cases"][case]
    ds = _dataset(case, d)
    prep = ToyPrep()
    bs = g["batch_size"]
    dl = data.DeviceLoader(ds, bs, prep, DEV, shuffle=False)
    dl16 = data.DeviceLoader(ds, bs, prep, DEV, shuffle=False, feat_dtype=torch.bfloat16)
    host = DataLoader(ds, batch_size=bs, collate_fn=data.collate_fn, shuffle=False)
    n = 0
    for i, ((f, m, caps, vids), (f16, m16, _c16, _v16), (hf, hm, hcaps, hvids)) in enumerate(zip(dl, dl16, host)):
        assert f[0].dtype == torch.float32 and m[0].dtype == torch.bool and f[0].is_cuda
        if not case.endswith("by_video"):            # by_video: directory order; the host loader of THIS build is the yardstick
            assert np.array_equal(f[0].cpu().numpy(), z[f"{case}.{i}.feat"])
            assert np.array_equal(m[0].cpu().numpy(), z[f"{case}.{i}.mask"])
            assert list(vids) == g["batches"][i]["vids"]
            want_ids, _ = prep(g["batches"][i]["captions"])
            assert torch.equal(caps.cpu(), want_ids)
        assert torch.equal(f[0].cpu(), hf[0]) and torch.equal(m[0].cpu(), hm[0]) and tuple(vids) == tuple(hvids)
        assert torch.equal(f16[0].cpu(), hf[0].to(torch.bfloat16)) and torch.equal(m16[0].cpu(), hm[0])
        n += 1
    assert n == len(g["batches"]) == len(dl)


def test_gather_pad_rows_edges():
    from vct_amd import ops
    rng = np.random.default_rng(0)
    for E in (10, 512, 4):                      # 10: scalar path (rows not 16-byte aligned)
        lens = [1, 7, 3, 12, 5]
        store = torch.from_numpy(rng.standard_normal((sum(lens), E)).astype(np.float32)).to(DEV)
        off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64, device=DEV)
        for idx in ([3], [0, 0, 4, 2], [1, 2, 3, 4, 0, 3]):
            tmax = max(lens[i] for i in idx)
            out, mask = ops.gather_pad_rows(store, off, torch.tensor(idx, device=DEV), tmax)
            for b, c in enumerate(idx):
                a = int(off[c])
                assert torch.equal(out[b, :lens[c]], store[a:a + lens[c]])
                assert float(out[b, lens[c]:].abs().sum()) == 0.0
                assert mask[b].tolist() == [False] * lens[c] + [True] * (tmax - lens[c])


def _steps(trainer, loader, n, start=0):
    losses = []
    it = [b for b in loader]
    for k in range(start, start + n):
        f, m, caps, _v = it[k % len(it)]
        losses.append(trainer.step(f[0], m[0], trainer.model.cap_preprocessor(caps)[0]))
    return torch.cat(losses)


def test_resume_is_bit_exact(split, tmp_path):
    """2 steps + save + load into a fresh model/optimizer + 2 steps == 4 uninterrupted steps, dropout 0.3 active:
    parameters, Adam moments, step counter, LR schedule and the device-side dropout seed all come back."""
    from vct_amd import checkpoint as ck, data
    from vct_amd.trainer import CaptionTrainer, build_optimizer
    _z, _meta, d = split
    ds = _dataset("msrvtt_train_by_caption", d)

    def fresh(seed):
        m = _model(torch.bfloat16, seed)
        m.train()
        opt, sch = build_optimizer(TC, m)
        return m, opt, sch, CaptionTrainer(m, opt), data.DeviceLoader(ds, 4, ToyPrep(), DEV, shuffle=True, seed=1)
    mA, optA, schA, trA, dlA = fresh(11)
    lossA = _steps(trA, dlA, 2)
    schA.step()
    lossA = torch.cat([lossA, _steps(trA, dlA, 2, start=2)])

    mB, optB, schB, trB, dlB = fresh(11)
    lossB = _steps(trB, dlB, 2)
    schB.step()
    ck.save_training_state(str(tmp_path / "s.pt"), mB, optB, schB, epoch=0)
    mC, optC, schC, trC, dlC = fresh(99)             # different init: everything must come from the file
    info = ck.load_training_state(str(tmp_path / "s.pt"), mC, optC, schC)
    assert info["epoch"] == 1 and _same_params(mC, mB)
    lossC = _steps(trC, dlC, 2, start=2)
    assert torch.equal(torch.cat([lossB, lossC]), lossA)
    assert _same_params(mC, mA)
    assert torch.equal(optC.exp_avg_sq, optA.exp_avg_sq) and int(optC.step_dev) == int(optA.step_dev) == 4
    assert optC.param_groups[0]["lr"] == optA.param_groups[0]["lr"] != TC["optimizer"]["learning_rate"]


def test_eval_and_val_epoch(split):
    from vct_amd import data, evaluate
    _z, meta, d = split
    m = _model(torch.float32, 5)
    ds = _dataset("msrvtt_val_by_video", d)
    dl = data.DeviceLoader(ds, 4, ToyPrep(), DEV, shuffle=False)
    res = evaluate.eval_epoch(m, dl, max_len=12)
    assert sorted(res) == sorted(meta["vids"]) and not any("[SEP]" in c or "[CLS]" in c for c in res.values())
    for f, k, _c, vids in dl:                       # same batches decoded directly
        direct = m.greedy_decode(f, k, max_len=12)
        assert [res[v] for v in vids] == [c.replace("[CLS]", "").replace("[SEP]", "") for c in direct]
    one = ds[2]
    assert evaluate.v2t_single(m, one[0], max_len=12) == evaluate.v2t_batch(m, [one[0][0][None]], None, max_len=12)[0]
    gts, samples, ids = evaluate.make_coco_sample(res, ds.video2caption)
    assert ids == list(res) and samples[ids[0]][0]["caption"] == res[ids[0]] and set(gts) == set(ds.video2caption)
    with pytest.raises(RuntimeError):
        evaluate.score_coco(gts, samples, ids)
    # validation loss: mean over batches of the eval-mode loss
    vds = _dataset("msrvtt_val_by_caption", d)
    vdl = data.DeviceLoader(vds, 3, ToyPrep(), DEV, shuffle=False)
    v = evaluate.val_epoch(m, vdl)
    m.eval()
    each = [float(m(f, k, caps).detach()) for f, k, caps, _v in vdl]
    assert abs(v - sum(each) / len(each)) < 1e-6 and np.isfinite(v)


def test_train_epoch_from_device_loader_learns(split):
    from vct_amd import data
    from vct_amd.trainer import build_optimizer, train_epoch
    _z, _meta, d = split
    ds = _dataset("msrvtt_train_by_caption", d)
    m = _model(torch.bfloat16, 7)
    opt, _sch = build_optimizer(TC, m)
    dl = data.DeviceLoader(ds, 8, ToyPrep(), DEV, shuffle=True, seed=3)
    first = train_epoch(m, opt, dl)
    for e in range(1, 12):
        dl.set_epoch(e)
        last = train_epoch(m, opt, dl)
    assert np.isfinite(first) and last < 0.85 * first        # 36 Adam steps on 22 captions: 8.4 -> ~6.5
