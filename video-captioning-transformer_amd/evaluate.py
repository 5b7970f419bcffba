"""Inference-side callers of the caption path (reference eval.py:126-168, train.py:171-203).

`v2t_batch` / `v2t_single` keep the reference signatures (minus `local_args`: the device is the model's).  `eval_epoch`
decodes a whole split in batches and returns {video id: caption}; `val_epoch` is the teacher-forced validation loss
(train.py:150-167).  COCO scoring (eval.py:42-123) needs Java + pycocoevalcap, neither of which this image has:
`make_coco_sample` reshapes the dictionaries exactly as eval.py:24-39 does so that an external scorer can be fed, and
`score_coco` says why it cannot run here instead of pretending."""
from typing import Dict, List, Optional, Sequence

import torch


def _strip_special(s: str) -> str:
    return s.replace("[CLS]", "").replace("[SEP]", "")          # eval.py:143, train.py:200


@torch.no_grad()
def v2t_batch(model, video_feats: Sequence[torch.Tensor], video_masks: Optional[Sequence[torch.Tensor]], max_len: int = 30) -> List[str]:
    """eval.py:126-145: video_feats = list (one per modality) of [B, T, E]; masks = list of bool [B, T] or None."""
    model.eval()
    dev = model.device
    video_feats = [f.to(dev, non_blocking=True) for f in video_feats]
    video_masks = [m.to(dev, non_blocking=True) for m in video_masks] if video_masks is not None else None
    return [_strip_special(r) for r in model.greedy_decode(video_feats, video_masks, max_len=max_len)]


@torch.no_grad()
def v2t_single(model, video_feat: Sequence[torch.Tensor], max_len: int = 30) -> str:
    """train.py:194-203: one video (list of [T, E] per modality), no mask."""
    model.eval()
    feats = [f.unsqueeze(0).to(model.device) for f in video_feat]
    return _strip_special(model.greedy_decode(feats, max_len=max_len)[0])


@torch.no_grad()
def eval_epoch(model, dataloader, max_len: int = 30) -> Dict[str, str]:
    """train.py:171-180 / eval.py:156-160 without the scorer: captions for every video a by_video loader yields
    -> {vid: caption}.  Decoding is batched (the reference's eval config uses batch_size 1)."""
    model.eval()
    vid2result: Dict[str, str] = {}
    for v_feats, v_masks, _caps, vids in dataloader:
        vid2result.update(zip(vids, v2t_batch(model, v_feats, v_masks, max_len=max_len)))
    return vid2result


@torch.no_grad()
def val_epoch(model, dataloader, mode: str = "caption") -> float:
    """train.py:150-167 for the caption task: mean teacher-forced loss over the loader, model in eval mode."""
    if mode != "caption":
        raise NotImplementedError("only the caption task is on the accelerated path")
    model.eval()
    model.mode(mode)
    dev = model.device
    total, n = torch.zeros(1, device=dev), 0
    for v_feats, v_masks, captions, _vids in dataloader:
        v_feats = [f.to(dev, non_blocking=True) for f in v_feats]
        v_masks = [m.to(dev, non_blocking=True) for m in v_masks]
        total += model(v_feats, v_masks, captions).detach().reshape(1)
        n += 1
    return float(total) / max(n, 1)


def make_coco_sample(prediction_dict: Dict[str, str], ground_truth_dict: Dict[str, List[str]]):
    """eval.py:24-39: (gts, samples, IDs) in the layout COCOScorer.score expects."""
    samples, IDs, gts = {}, [], {}
    for vid, cap in prediction_dict.items():
        IDs.append(vid)
        samples[vid] = [{u"image_id": vid, u"caption": cap}]
    for vid, caps in ground_truth_dict.items():
        gts[vid] = [{u"image_id": vid, u"caption": cap} for cap in caps]
    return gts, samples, IDs


def score_coco(gts, samples, IDs):
    """BLEU/METEOR/ROUGE/CIDEr (eval.py:42-123) shell out to Java through pycocoevalcap.  Not available in this image."""
    try:
        from pycocoevalcap.bleu.bleu import Bleu      # noqa: F401
    except Exception as e:
        raise RuntimeError("COCO caption scoring needs pycocoevalcap + a Java runtime (reference submodule "
                           "submodules/pycocoevalcap); neither is installed here. Feed make_coco_sample()'s output to "
                           "the reference's COCOScorer on a machine that has them.") from e
    raise RuntimeError("pycocoevalcap found but scoring is not wired: use the reference's COCOScorer")
