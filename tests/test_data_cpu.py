"""Input path, early stopping and checkpoint files against fixtures recorded from the reference's own
dataloader.py / utils.py (oracle/make_golden_data.py) -- CPU only."""
import json
import os

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

import vct_oracle as O
from helpers import GOLDEN, build_model, load_golden


@pytest.fixture(scope="module")
def split(tmp_path_factory):
    """The synthetic split of the fixture, rebuilt on disk in the reference's formats."""
    z = load_golden("dataloader.npz")
    meta = json.loads(str(z["meta"]))
    d = tmp_path_factory.mktemp("split")
    os.makedirs(d / "feats")
    for v in meta["vids"]:
        np.save(d / "feats" / f"{v}.npy", z[f"clip_{v}"])
    (d / "ann.json").write_text(json.dumps(meta["annotation"]))
    (d / "msvd_train.txt").write_text(meta["msvd_train_txt"])
    return z, meta, d


def _dataset(case, d):
    from vct_amd import data
    fd = [str(d / "feats")]
    if case.startswith("msvd"):
        return data.MSVD_Dataset(fd, str(d / "msvd_train.txt"), split_type="train", mode="by_caption")
    split_type = "train" if "train" in case else "val"
    mode = "by_video" if case.endswith("by_video") else "by_caption"
    kw = dict(debug=True, debug_num=5) if case.endswith("debug") else {}
    return data.MSRVTT_Dataset(fd, str(d / "ann.json"), split_type=split_type, mode=mode, **kw)


CASES = ["msrvtt_train_by_caption", "msrvtt_val_by_caption", "msrvtt_val_by_video", "msrvtt_train_debug", "msvd_train_by_caption"]


@pytest.mark.parametrize("case", CASES)
def test_dataset_and_collate_match_reference(split, case):
    from vct_amd import data
    z, meta, d = split
    g = meta["cases"][case]
    ds = _dataset(case, d)
    assert len(ds) == g["len"]
    assert [(c, p[0].stem) for c, p in ds.cap_vid_list] == [tuple(x) for x in g["cap_vid_list"]]
    assert ds.video2caption == g["video2caption"]
    dl = DataLoader(ds, batch_size=g["batch_size"], collate_fn=data.collate_fn, shuffle=False)
    if case.endswith("by_video"):
        # by_video walks the directory listing, whose order is the file system's: compare per video
        def per_video(batches):
            out = {}
            for feat, mask, vids in batches:
                for i, v in enumerate(vids):
                    out[v] = np.asarray(feat[i])[~np.asarray(mask[i])]
            return out
        mine = per_video((f[0].numpy(), m[0].numpy(), vids) for f, m, _c, vids in dl)
        ref = per_video((z[f"{case}.{i}.feat"], z[f"{case}.{i}.mask"], b["vids"]) for i, b in enumerate(g["batches"]))
        assert mine.keys() == ref.keys()
        assert all(np.array_equal(mine[v], ref[v]) for v in ref)
        return
    n = 0
    for i, (feats, masks, caps, vids) in enumerate(dl):
        assert len(feats) == 1 and feats[0].dtype == torch.float32 and masks[0].dtype == torch.bool
        assert np.array_equal(feats[0].numpy(), z[f"{case}.{i}.feat"])          # bit-exact: it is a copy
        assert np.array_equal(masks[0].numpy(), z[f"{case}.{i}.mask"])
        assert list(caps) == g["batches"][i]["captions"] and list(vids) == g["batches"][i]["vids"]
        n += 1
    assert n == len(g["batches"])


def test_oracle_collate_matches_reference(split):
    z, meta, _d = split
    case = "msrvtt_train_by_caption"
    for i, b in enumerate(meta["cases"][case]["batches"]):
        feat, mask = O.make_mask_video([O.load_clip(z[f"clip_{v}"]) for v in b["vids"]])
        assert np.array_equal(feat, z[f"{case}.{i}.feat"]) and np.array_equal(mask, z[f"{case}.{i}.mask"])


def test_early_stopping_matches_reference(tmp_path):
    from vct_amd.checkpoint import EarlyStopping
    g = json.load(open(os.path.join(GOLDEN, "early_stopping.json")))

    class Rec:
        saved = 0

        def state_dict(self):
            self.saved += 1
            return {}
    for name, c in g.items():
        es = EarlyStopping(patience=c["patience"], delta=c["delta"], path=str(tmp_path / "m.pt"), trace_func=lambda *_a: None)
        rec = Rec()
        oracle = O.early_stopping_trace(c["losses"], c["patience"], c["delta"])
        for v, want, ow in zip(c["losses"], c["trace"], oracle):
            es(v, rec, do_save=True)
            got = {"counter": es.counter, "best_score": es.best_score, "early_stop": es.early_stop,
                   "val_loss_min": float(es.val_loss_min), "saves": rec.saved}
            assert got == want, (name, v, got, want)
            assert ow == want, (name, v, ow, want)
        # counters survive a state_dict round trip
        es2 = EarlyStopping(patience=c["patience"], delta=c["delta"])
        es2.load_state_dict(es.state_dict())
        assert (es2.counter, es2.best_score, es2.early_stop) == (es.counter, es.best_score, es.early_stop)


class ToyTok:
    """Whitespace tokenizer standing in for bert-base-uncased (no vocabulary files offline)."""
    vocab_size = 2000
    _special = {"[PAD]": 0, "[CLS]": 101, "[SEP]": 102}

    def convert_tokens_to_ids(self, tok):
        return self._special[tok]

    def encode(self, text, **_):
        return [101] + [1000 + (sum(map(ord, w)) % 500) for w in text.split()] + [102]

    def convert_ids_to_tokens(self, ids):
        inv = {v: k for k, v in self._special.items()}
        return [inv.get(int(i), f"w{int(i)}") for i in ids]

    def convert_tokens_to_string(self, toks):
        return " ".join(toks)


class ToyPrep:
    """CapPreprocessor-shaped callable over ToyTok (host tensors)."""
    pad_id, start_id, end_id = 0, 101, 102
    tokenizer = ToyTok()

    def __call__(self, captions):
        rows = [self.tokenizer.encode(c) for c in captions]
        S = max(map(len, rows))
        ids = torch.tensor([r + [0] * (S - len(r)) for r in rows], dtype=torch.long)
        return ids, ids == 0


@pytest.mark.parametrize("world", [1, 2, 3])
def test_device_loader_sampling_is_distributed_sampler(split, world):
    from vct_amd import data
    _z, _meta, d = split
    ds = _dataset("msrvtt_train_by_caption", d)
    for epoch in (0, 1, 5):
        seen = []
        for rank in range(world):
            dl = data.DeviceLoader(ds, 4, ToyPrep(), "cpu", shuffle=True, rank=rank, world=world, seed=0)
            dl.set_epoch(epoch)
            idx = dl._indices()
            if world > 1:
                s = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=0)
                s.set_epoch(epoch)
                assert list(idx) == list(iter(s))
            else:
                g = torch.Generator()
                g.manual_seed(epoch)
                assert list(idx) == torch.randperm(len(ds), generator=g).tolist()
            seen += list(idx)
        assert set(seen) == set(range(len(ds)))
    seq = data.DeviceLoader(ds, 4, ToyPrep(), "cpu", shuffle=False)
    assert list(seq._indices()) == list(range(len(ds))) and len(seq) == (len(ds) + 3) // 4


def _same_params(a, b):
    """named parameters only: the flat buffers also hold alignment padding, which no checkpoint stores"""
    sa, sb = a.state_dict(), b.state_dict()
    return sa.keys() == sb.keys() and all(torch.equal(sa[k], sb[k]) for k in sa)


def test_weight_file_and_training_state_round_trip(tmp_path):
    from vct_amd import checkpoint as ck
    from vct_amd.trainer import build_optimizer
    mc = {"modal": ["x"], "modal_shape": [24], "text_enc_type": "CLIP", "embed_dim": 32, "dropout": 0.1, "loss_beta": 0.5,
          "matching": {"enable_tem": False, "matching_loss": "CSL"}, "activation": "gelu",
          "video_encoder": {"layer": 1, "nhead": 4, "feedforward": 48,
                            "mme": {"temporal": "encoding", "modal_different": True, "do_norm": False, "aggregation": "avg"}, "aoa": False},
          "caption_decoder": {"layer": 1, "nhead": 4, "feedforward": 48, "sce_loss_alpha": 0.5}, "pretrained_model": None}
    tc = {"optimizer": {"name": "adam", "learning_rate": 1e-3, "beta": [0.9, 0.999], "weight_decay": 0,
                        "lr_scheduler": {"name": "CosineAnnealingLR", "T_max": 10, "eta_min": 1e-5}}}
    torch.manual_seed(3)
    m1 = build_model(mc, 97, "cpu", torch.float32)
    opt1, sch1 = build_optimizer(tc, m1)
    for _ in range(3):                      # three optimizer steps on made-up gradients (the kernels need a GPU)
        m1.flat_grads.copy_(torch.randn_like(m1.flat_grads))
        opt1.step()
        sch1.step()
    es1 = ck.EarlyStopping(patience=4)
    es1(2.0, m1, do_save=False)
    es1(2.5, m1, do_save=False)
    # reference-format weight file: a bare state_dict with the reference's key names
    ck.save_weights(m1, str(tmp_path / "w.pth"))
    sd = torch.load(str(tmp_path / "w.pth"))
    assert "cap_decoder.generator.weight" in sd and all(v.dtype == torch.float32 for v in sd.values() if v.is_floating_point())
    torch.manual_seed(4)
    m2 = build_model(mc, 97, "cpu", torch.float32)
    assert not _same_params(m2, m1)
    res = ck.load_weights(m2, str(tmp_path / "w.pth"))
    assert not res.unexpected_keys and _same_params(m2, m1)
    # full training state
    ck.save_training_state(str(tmp_path / "s.pt"), m1, opt1, sch1, epoch=2, early_stopping=es1, extra={"tag": "t"})
    torch.manual_seed(5)
    m3 = build_model(mc, 97, "cpu", torch.float32)
    opt3, sch3 = build_optimizer(tc, m3)
    es3 = ck.EarlyStopping(patience=4)
    info = ck.load_training_state(str(tmp_path / "s.pt"), m3, opt3, sch3, es3)
    assert info == {"epoch": 3, "extra": {"tag": "t"}}
    assert _same_params(m3, m1)
    assert sch3.state_dict() == sch1.state_dict() and opt3.param_groups[0]["lr"] == opt1.param_groups[0]["lr"]
    assert (es3.counter, es3.best_score) == (1, -2.0)
    g = torch.randn_like(m1.flat_grads)     # the next step is identical on both sides
    for m, o in ((m1, opt1), (m3, opt3)):
        m.flat_grads.copy_(g)
        o.step()
    assert _same_params(m3, m1)
    with pytest.raises(ValueError):
        ck.load_training_state(str(tmp_path / "w.pth"), m3)       # a bare weight file is not a training state


def test_training_state_with_numpy_scalars_loads_again(tmp_path):
    """Validation losses usually arrive as numpy scalars (np.mean of per-batch losses): EarlyStopping.best_score and
    ReduceLROnPlateau.best then hold np.float64, which torch.save pickles happily and a weights_only load refuses.  The file
    must be readable: scalars are stored as plain Python numbers."""
    from vct_amd import checkpoint as ck
    from vct_amd.trainer import build_optimizer
    mc = {"modal": ["x"], "modal_shape": [24], "text_enc_type": "CLIP", "embed_dim": 32, "dropout": 0.1, "loss_beta": 0.5,
          "matching": {"enable_tem": False, "matching_loss": "CSL"}, "activation": "gelu",
          "video_encoder": {"layer": 1, "nhead": 4, "feedforward": 48,
                            "mme": {"temporal": "encoding", "modal_different": True, "do_norm": False, "aggregation": "avg"}, "aoa": False},
          "caption_decoder": {"layer": 1, "nhead": 4, "feedforward": 48, "sce_loss_alpha": 0.5}, "pretrained_model": None}
    tc = {"optimizer": {"name": "adam", "learning_rate": 1e-3, "beta": [0.9, 0.999], "weight_decay": 0,
                        "lr_scheduler": {"name": "ReduceLROnPlateau", "patience": 2}}}
    m = build_model(mc, 97, "cpu", torch.float32)
    opt, sch = build_optimizer(tc, m)
    es = ck.EarlyStopping(patience=3)
    for v in (np.float64(2.25), np.mean([2.0, 3.0]), np.float32(2.75)):
        sch.step(v)
        es(v, m, do_save=False)
    assert isinstance(es.best_score, np.floating)        # the hazard is real: utils.py:38 keeps whatever type it was given
    path = str(tmp_path / "s.pt")
    ck.save_training_state(path, m, opt, sch, epoch=0, early_stopping=es, extra={"val": np.float64(1.5), "n": np.int64(7)})
    m2 = build_model(mc, 97, "cpu", torch.float32)
    opt2, sch2 = build_optimizer(tc, m2)
    es2 = ck.EarlyStopping(patience=3)
    info = ck.load_training_state(path, m2, opt2, sch2, es2)
    assert info["extra"] == {"val": 1.5, "n": 7} and type(info["extra"]["val"]) is float
    assert sch2.best == 2.25 and sch2.num_bad_epochs == sch.num_bad_epochs
    # (val_loss_min holds the NEGATED value: the reference's own bookkeeping, utils.py:38-47)
    assert (es2.counter, es2.best_score, es2.val_loss_min) == (es.counter, -2.25, float(es.val_loss_min)) and type(es2.best_score) is float
