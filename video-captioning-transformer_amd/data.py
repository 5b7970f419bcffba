"""Input side of the caption path (reference dataloader.py:233-247, 340-540; train.py:119-121; CapPreprocessor.py:24-36).

Two ways to feed the model, same batches:

* the reference's own layout, for drop-in use: `MSRVTT_Dataset` / `MSVD_Dataset` items `(list of [T,E] fp32 per modality,
  caption str, video id)`, `collate_fn` -> `(list of [B,Tmax,E], list of bool [B,Tmax] (True = padded), captions, vids)`,
  `build_dataloader(data_cfg, multi_gpu)` -> `(dataset, DataLoader, sampler)` with the reference's config keys.

* `DeviceLoader`, the MI355X-native path.  A split's precomputed features are a few hundred MB (MSR-VTT train: 6513 videos
  x 12 x 512 fp32 = 160 MB) and the GPU has 288 GB: the whole split is uploaded ONCE into a packed [rows, E] buffer, the
  captions are tokenised ONCE into an id matrix, and a batch is one gather-and-pad kernel launch (vct_gather_pad_rows)
  plus one index_select -- no per-sample np.load, no per-step H2D copy, no host tokeniser in the step loop (the
  reference does B np.load + B tokenizer.encode + 2B device round trips per step, which caps it near 1-2 k samples/s).
  Sampling is the reference's: `torch.randperm` seeded by `seed + epoch`, padded to a multiple of the world size and
  strided by rank exactly like `DistributedSampler(shuffle=True)`, or sequential when shuffle is off.
"""
import json
import pathlib as plb
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler


def _load_feat(path) -> np.ndarray:
    """dataloader.py:378-386: fp32, and [E, T] files are turned into [T, E]."""
    a = np.load(str(path)).astype(np.float32, copy=False)
    return np.ascontiguousarray(a.T) if a.shape[0] > a.shape[1] else a


def _make_mask_video(ts: Sequence[torch.Tensor]):
    """dataloader.py:233-247: zero-pad to the longest clip of the batch; mask True = padded frame."""
    B, E = len(ts), ts[0].shape[1]
    lens = [t.shape[0] for t in ts]
    feat = torch.zeros(B, max(lens), E, dtype=torch.float32)
    mask = torch.ones(B, max(lens), dtype=torch.bool)
    for i, t in enumerate(ts):
        feat[i, :lens[i]] = t
        mask[i, :lens[i]] = False
    return feat, mask


def collate_fn(data):
    """dataloader.py:507-510: -> (list[M] of [B,T,E], list[M] of [B,T] bool, captions tuple, vids tuple)."""
    batch_feats, batch_captions, batch_vids = list(zip(*data))
    feats, masks = [], []
    for modal in zip(*batch_feats):            # dataloader.py:265-274
        f, m = _make_mask_video(modal)
        feats.append(f)
        masks.append(m)
    return feats, masks, batch_captions, batch_vids


class CaptionDataset(Dataset):
    """Shared part of MSRVTT_Dataset / MSVD_Dataset (dataloader.py:340-386)."""

    def __init__(self, video_feat_dirs: List[str], annotation_file: str, split_type="train", mode: str = "by_caption",
                 debug: bool = False, debug_num: int = 400):
        if split_type.lower() in ("val", "validate"):
            split_type = "validate"
        self.split_type, self.mode = split_type, mode
        self.annotation_file, self.video_feat_dirs = annotation_file, video_feat_dirs
        per_dir = [sorted(plb.Path(d).glob("*.npy")) for d in video_feat_dirs]
        self.video_feat_list: List[Tuple[plb.Path, ...]] = list(zip(*per_dir))
        self.cap_vid_list, self.video2caption = self.make_cap_vid_list()
        if debug is True:
            self.cap_vid_list = self.cap_vid_list[:debug_num]

    def make_cap_vid_list(self):
        raise NotImplementedError

    def _pairs(self, video2caption):
        video2path = {p[0].stem: p for p in self.video_feat_list}
        return [(cap, video2path[vid]) for vid, caps in video2caption.items() for cap in caps]

    def __len__(self):
        if self.mode == "by_caption":
            return len(self.cap_vid_list)
        if self.mode == "by_video":
            return len(self.video_feat_list)
        raise ValueError

    def __getitem__(self, index):
        if self.mode == "by_caption":
            caption, paths = self.cap_vid_list[index]
        elif self.mode == "by_video":
            caption, paths = "", self.video_feat_list[index]
        else:
            raise ValueError
        return [torch.from_numpy(_load_feat(p)) for p in paths], caption, paths[0].stem


class MSRVTT_Dataset(CaptionDataset):
    """dataloader.py:399-454: train_val_videodatainfo.json -- {"videos": [{video_id, split}], "sentences":
    [{video_id, caption}]}; keeps the sentences of videos whose split equals split_type."""

    def make_cap_vid_list(self):
        with open(self.annotation_file, encoding="utf-8") as f:
            ann = json.load(f)
        split = {v["video_id"]: v["split"] for v in ann["videos"]}
        video2caption: Dict[str, List[str]] = {}
        for s in ann["sentences"]:
            if split[s["video_id"]] == self.split_type:
                video2caption.setdefault(s["video_id"], []).append(s["caption"])
        return self._pairs(video2caption), video2caption


class MSVD_Dataset(CaptionDataset):
    """dataloader.py:457-504: one '<vid> <caption words...>' line per caption; the file IS the split."""

    def make_cap_vid_list(self):
        video2caption: Dict[str, List[str]] = {}
        with open(self.annotation_file) as f:
            for line in f.readlines():
                parts = line.split(" ")
                video2caption.setdefault(parts[0], []).append(" ".join(parts[1:]).replace("\n", ""))
        return self._pairs(video2caption), video2caption


def build_dataset(data_cfg: dict) -> CaptionDataset:
    cls = MSRVTT_Dataset if data_cfg.get("dataset", "msrvtt") == "msrvtt" else MSVD_Dataset
    return cls(data_cfg["feat_dir"], data_cfg["annotation_path"], split_type=data_cfg["split_mode"], mode=data_cfg["mode"],
               debug=data_cfg.get("_debug", False), debug_num=data_cfg.get("_debug_num", 400))


def build_dataloader(data_cfg: dict, multi_gpu: bool):
    """dataloader.py:513-533 (host path; pinned memory added so the .to(device, non_blocking=True) copies overlap)."""
    ds = build_dataset(data_cfg)
    sampler = DistributedSampler(ds, shuffle=True) if (data_cfg["split_mode"] == "train" and multi_gpu) else None
    dl = DataLoader(ds, batch_size=data_cfg["batch_size"], collate_fn=collate_fn, sampler=sampler,
                    shuffle=(data_cfg["split_mode"] == "train" and not multi_gpu), pin_memory=torch.cuda.is_available())
    return ds, dl, sampler


# ------------------------------------------------------------------------------------------------
class FeatureStore:
    """All clips of one modality of a split, packed: data [rows, E] fp32, offsets int64 [n+1]; clip i = rows
    offsets[i]:offsets[i+1].  `to(device)` makes it HBM-resident."""

    def __init__(self, paths: Sequence):
        clips = [_load_feat(p) for p in paths]
        self.stems = [plb.Path(p).stem for p in paths]
        self.lens = np.array([c.shape[0] for c in clips], dtype=np.int64)
        self.offsets_host = np.concatenate([[0], np.cumsum(self.lens)]).astype(np.int64)
        self.E = clips[0].shape[1]
        self.data = torch.from_numpy(np.concatenate(clips, 0)) if clips else torch.zeros(0, 0)
        self.offsets = torch.from_numpy(self.offsets_host)

    def to(self, device):
        self.data, self.offsets = self.data.to(device), self.offsets.to(device)
        return self

    def gather(self, idx_host: np.ndarray, idx_dev: torch.Tensor, out_dtype=torch.float32):
        """-> (feat [B, Tmax, E], mask bool [B, Tmax]) for clips idx; Tmax = longest clip of the batch (known on the
        host: no sync)."""
        from . import ops
        tmax = int(self.lens[idx_host].max())
        return ops.gather_pad_rows(self.data, self.offsets, idx_dev, tmax, out_dtype)


class DeviceLoader:
    """Iterates a CaptionDataset with everything resident on the device (see the module docstring).  Yields the
    reference's batch tuple `(v_feats, v_masks, captions, vids)`; `captions` is an int64 id tensor [B, S] (pads =
    pad_id, trimmed to the longest caption of the batch), which MMT4Caption.forward accepts in place of strings."""

    def __init__(self, dataset: CaptionDataset, batch_size: int, preprocessor, device, shuffle: bool = False,
                 rank: int = 0, world: int = 1, seed: int = 0, drop_last: bool = False, feat_dtype=torch.float32):
        if len(dataset.video_feat_dirs) != 1:
            raise NotImplementedError("the accelerated caption path takes one modality (MMEncoder with a single feature stream)")
        self.ds, self.bs, self.device, self.shuffle = dataset, batch_size, torch.device(device), shuffle
        self.rank, self.world, self.seed, self.drop_last, self.epoch = rank, world, seed, drop_last, 0
        self.feat_dtype = feat_dtype
        paths = [p[0] for p in dataset.video_feat_list]
        self.store = FeatureStore(paths).to(self.device)
        row = {s: i for i, s in enumerate(self.store.stems)}
        if dataset.mode == "by_caption":
            self.clip_of_item = np.array([row[p[0].stem] for _c, p in dataset.cap_vid_list], dtype=np.int64)
            ids, mask = preprocessor([c for c, _p in dataset.cap_vid_list])       # tokenised ONCE
            self.ids = ids.to(self.device)
            self.cap_len = (~mask).sum(1).cpu().numpy().astype(np.int64)
            self.pad_id = preprocessor.pad_id
        else:
            self.clip_of_item = np.arange(len(paths), dtype=np.int64)
            self.ids, self.cap_len = None, None
        self.clip_of_item_dev = torch.from_numpy(self.clip_of_item).to(self.device)

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def _indices(self) -> np.ndarray:
        n = len(self.clip_of_item)
        if self.shuffle:        # DistributedSampler.__iter__ (shuffle=True): randperm seeded with seed + epoch
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(n, generator=g).numpy()
        else:
            order = np.arange(n)
        if self.world > 1:      # pad by wrapping to a multiple of world, then stride by rank
            total = (n + self.world - 1) // self.world * self.world
            if total > n:
                order = np.concatenate([order, np.resize(order, total - n)])
            order = order[self.rank:total:self.world]
        return order

    def __len__(self):
        n = len(self._indices())
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def __iter__(self):
        order = self._indices()
        order_dev = torch.from_numpy(order).to(self.device, non_blocking=True)
        stop = len(order) - (len(order) % self.bs if self.drop_last else 0)
        for a in range(0, stop, self.bs):
            item_h, item_d = order[a:a + self.bs], order_dev[a:a + self.bs]
            clip_h = self.clip_of_item[item_h]
            clip_d = self.clip_of_item_dev.index_select(0, item_d)
            feat, mask = self.store.gather(clip_h, clip_d, self.feat_dtype)
            if self.ids is not None:
                S = int(self.cap_len[item_h].max())
                caps = self.ids.index_select(0, item_d)[:, :S]
            else:
                caps = tuple("" for _ in item_h)
            yield [feat], [mask], caps, tuple(self.store.stems[i] for i in clip_h)
