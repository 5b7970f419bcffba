"""Generate tests/golden/dataloader.npz and early_stopping.json from the REAL reference's input path and
early-stopping rule (runs only in the build container: needs /root/reference).

    python oracle/make_golden_data.py

A small synthetic split is written to a temporary directory in the two on-disk formats the reference reads
(MSR-VTT `train_val_videodatainfo.json`, MSVD `<vid> <caption>` lines; one `.npy` per video, ragged lengths, one of
them stored [E, T]); the reference's own `MSRVTT_Dataset` / `MSVD_Dataset` / `collate_fn` / `DataLoader` then produce
the batches that are recorded.  The synthetic inputs travel inside the fixture, so the tests rebuild the same files.
Nothing from the reference is copied: inputs and outputs only.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("VCT_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
if not hasattr(np, "Inf"):
    np.Inf = np.inf          # the reference (utils.py:32) predates numpy 2

import dataloader as RD  # noqa: E402  (the reference)
from utils import EarlyStopping as RefEarlyStopping  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def synth_split(rng):
    vids = [f"video{i}" for i in range(9)]
    E = 16
    clips = {}
    for i, v in enumerate(vids):
        T = int(rng.integers(3, 13))
        a = rng.standard_normal((T, E)).astype(np.float32)
        clips[v] = np.ascontiguousarray(a.T) if i == 4 else a          # video4 is stored [E, T]
    split = {v: ("train" if i % 3 != 2 else "validate") for i, v in enumerate(vids)}
    words = "a man woman dog is are playing cooking running with the ball guitar in kitchen park".split()
    sentences = []
    for k in range(23):
        v = vids[int(rng.integers(0, len(vids)))]
        n = int(rng.integers(3, 9))
        sentences.append({"video_id": v, "caption": " ".join(words[int(j)] for j in rng.integers(0, len(words), n))})
    for v in vids:                                                       # every video has at least one caption
        sentences.append({"video_id": v, "caption": f"{v} something happens"})
    ann = {"videos": [{"video_id": v, "split": split[v]} for v in vids], "sentences": sentences}
    msvd = "".join(f"{s['video_id']} {s['caption']}\n" for s in sentences if split[s["video_id"]] == "train")
    return vids, clips, ann, msvd


def write_split(d, clips, ann, msvd):
    os.makedirs(os.path.join(d, "feats"))
    for v, a in clips.items():
        np.save(os.path.join(d, "feats", v + ".npy"), a)
    with open(os.path.join(d, "ann.json"), "w") as f:
        json.dump(ann, f)
    with open(os.path.join(d, "msvd_train.txt"), "w") as f:
        f.write(msvd)


def record(ds, bs):
    dl = RD.DataLoader(ds, batch_size=bs, collate_fn=RD.collate_fn, shuffle=False)
    batches = []
    for feats, masks, caps, vids in dl:
        batches.append({"feat": feats[0].numpy(), "mask": masks[0].numpy(), "captions": list(caps), "vids": list(vids)})
    return batches


def main():
    rng = np.random.default_rng(20240917)
    vids, clips, ann, msvd = synth_split(rng)
    arrays, meta = {f"clip_{v}": a for v, a in clips.items()}, {"vids": vids, "annotation": ann, "msvd_train_txt": msvd, "cases": {}}
    with tempfile.TemporaryDirectory() as d:
        write_split(d, clips, ann, msvd)
        fd, aj, at = [os.path.join(d, "feats")], os.path.join(d, "ann.json"), os.path.join(d, "msvd_train.txt")
        cases = {
            "msrvtt_train_by_caption": (RD.MSRVTT_Dataset(fd, aj, split_type="train", mode="by_caption"), 4),
            "msrvtt_val_by_caption": (RD.MSRVTT_Dataset(fd, aj, split_type="val", mode="by_caption"), 3),
            "msrvtt_val_by_video": (RD.MSRVTT_Dataset(fd, aj, split_type="validate", mode="by_video"), 2),
            "msrvtt_train_debug": (RD.MSRVTT_Dataset(fd, aj, split_type="train", mode="by_caption", debug=True, debug_num=5), 5),
            "msvd_train_by_caption": (RD.MSVD_Dataset(fd, at, split_type="train", mode="by_caption"), 4),
        }
        for name, (ds, bs) in cases.items():
            batches = record(ds, bs)
            for i, b in enumerate(batches):
                arrays[f"{name}.{i}.feat"], arrays[f"{name}.{i}.mask"] = b["feat"], b["mask"]
            meta["cases"][name] = {
                "batch_size": bs, "len": len(ds),
                "cap_vid_list": [(c, p[0].stem) for c, p in ds.cap_vid_list],
                "video2caption": ds.video2caption,
                "batches": [{"captions": b["captions"], "vids": b["vids"]} for b in batches],
            }
    np.savez_compressed(os.path.join(OUT, "dataloader.npz"), meta=json.dumps(meta), **arrays)

    # early stopping: the reference's counters after each call of a loss sequence, and which calls saved
    class Rec:
        def __init__(self):
            self.saved = 0

        def state_dict(self):
            self.saved += 1
            return {}
    seqs = {"improve_then_stall": [5.0, 4.0, 4.5, 4.2, 3.9, 4.0, 4.0, 4.0], "delta": [1.0, 0.95, 0.97, 0.80, 0.85, 0.86, 0.87],
            "negated_metric": [-0.30, -0.35, -0.33, -0.36, -0.10, -0.10, -0.10]}
    es_out = {}
    with tempfile.TemporaryDirectory() as d:
        for name, seq in seqs.items():
            es = RefEarlyStopping(patience=3, verbose=False, delta=0.02 if name == "delta" else 0, path=os.path.join(d, "m.pt"),
                                  trace_func=lambda *_a: None)
            rec, trace = Rec(), []
            for v in seq:
                es(v, rec, do_save=True)
                trace.append({"counter": es.counter, "best_score": es.best_score, "early_stop": es.early_stop,
                              "val_loss_min": float(es.val_loss_min), "saves": rec.saved})
            es_out[name] = {"patience": 3, "delta": 0.02 if name == "delta" else 0, "losses": seq, "trace": trace}
    with open(os.path.join(OUT, "early_stopping.json"), "w") as f:
        json.dump(es_out, f, indent=1)
    print("wrote dataloader.npz, early_stopping.json:", {k: v["len"] for k, v in meta["cases"].items()})


if __name__ == "__main__":
    main()
