"""Host-side logic that needs no GPU: flat gradient-ready parameter storage, state_dict surface of
the shipped config shape, preprocessing, mode flags, config/masks."""
import json
import os

import numpy as np
import torch

import vct_oracle as O
from helpers import build_model

SHIPPED_LIKE = {   # the values of the reference's shipped MSR-VTT JSON "model" block (config:63-95)
    "modal": ["CLIP4Clip"], "modal_shape": [512], "tokenizer": "bert-base-uncased", "text_enc_type": "CLIP",
    "embed_dim": 768, "dropout": 0.3, "loss_beta": 0.5, "matching": {"enable_tem": False, "matching_loss": "CSL"},
    "activation": "gelu",
    "video_encoder": {"layer": 1, "nhead": 8, "feedforward": 2048,
                      "mme": {"temporal": "encoding", "modal_different": True, "do_norm": False, "aggregation": "avg"}, "aoa": False},
    "caption_decoder": {"layer": 3, "nhead": 8, "feedforward": 2048, "sce_loss_alpha": 0.5},
    "pretrained_model": None}


def test_shipped_config_shape_param_count_and_keys():
    from vct_amd.model import MMT4Caption
    m = MMT4Caption(SHIPPED_LIKE, device=torch.device("cpu"), compute_dtype=torch.float32)
    m.mode("caption")
    n = sum(p.numel() for p in m.parameters())
    # SURVEY.md Appendix B counts 76 457 018 for encoder+decoder; the reference also owns matching.v_proj
    # (Linear(768 -> CLIP dim 512), Matching.py:21 with TextEncoder.dim = 512, TextEncoder.py:15): +393 728
    assert n == 76457018 + 393728
    sd = m.state_dict()
    assert "matching.v_proj.weight" in sd and tuple(sd["matching.v_proj.weight"].shape) == (512, 768)
    assert tuple(sd["cap_decoder.generator.weight"].shape) == (30522, 768)
    assert tuple(sd["video_encoder.transformer_encoder.layers.0.self_attn.in_proj_weight"].shape) == (2304, 768)
    assert "cap_decoder.decoder.layers.2.multihead_attn.out_proj.bias" in sd
    assert tuple(sd["cap_decoder.positional_encoding.pos_embedding"].shape) == (5000, 768)
    assert tuple(sd["video_encoder.temp_emb.pe"].shape) == (1, 512, 768)
    # mode(): caption -> decoder trainable, matching frozen, encoder untouched (MMT4Caption.py:189-211)
    assert all(p.requires_grad for p in m.cap_decoder.parameters())
    assert not any(p.requires_grad for p in m.matching.parameters())
    assert all(p.requires_grad for p in m.video_encoder.parameters())
    m.mode("match")
    assert not any(p.requires_grad for p in m.cap_decoder.parameters())
    try:
        m.mode("bogus"); assert False
    except ValueError:
        pass


def test_flat_storage_views_order_and_buckets():
    mc = dict(SHIPPED_LIKE, embed_dim=64, modal_shape=[48])
    mc["video_encoder"] = dict(mc["video_encoder"], layer=2, nhead=4, feedforward=128)
    mc["caption_decoder"] = dict(mc["caption_decoder"], layer=2, nhead=4, feedforward=128)
    m = build_model(mc, 131, "cpu", torch.float32)
    ps = m._ps
    assert ps.intact()
    # every parameter is a view of the flat buffer; writing the flat buffer changes the parameter
    for n, p in m.named_parameters():
        o = ps.offsets[n]
        assert p.data_ptr() == ps.flat.data_ptr() + 4 * o
        assert o % 64 == 0
    w = dict(m.named_parameters())["cap_decoder.generator.weight"]
    ps.flat[ps.offsets["cap_decoder.generator.weight"]] = 123.0
    assert float(w.view(-1)[0]) == 123.0
    # gradient-ready order: generator first, then decoder top-down, embedding, encoder, unify last
    names = ps.names
    assert names[0] == "cap_decoder.generator.weight"
    assert names.index("cap_decoder.decoder.layers.1.linear2.weight") < names.index("cap_decoder.decoder.layers.0.linear2.weight")
    assert names.index("cap_decoder.tgt_to_emb.weight") < names.index("video_encoder.transformer_encoder.norm.weight")
    assert names.index("video_encoder.unify.0.weight") > names.index("video_encoder.transformer_encoder.layers.0.self_attn.in_proj_weight")
    # buckets: contiguous, ordered, cover the whole flat gradient buffer
    b = m.grad_buckets()
    assert b[0][0] == 0 and b[-1][1] == ps.total
    assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
    assert len(b) == 1 + 2 + 1 + 2     # generator | 2 decoder layers | embedding | 2 encoder layers
    assert b[m.bucket_index("embedding")] == (ps.offsets["cap_decoder.tgt_to_emb.weight"], ps.offsets["video_encoder.transformer_encoder.norm.weight"])
    assert b[m.bucket_index("generator")] == (0, ps.offsets["cap_decoder.decoder.norm.weight"])
    a1, e1 = b[m.bucket_index("dec_layer", 1)]
    assert a1 == ps.offsets["cap_decoder.decoder.norm.weight"] and e1 == ps.offsets["cap_decoder.decoder.layers.0.norm3.weight"]
    a0, e0 = b[m.bucket_index("enc_layer", 0)]
    assert a0 == ps.offsets["video_encoder.transformer_encoder.layers.0.norm2.weight"] and e0 == ps.total
    # state_dict round trip keeps the aliasing (load_state_dict copies in place)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["cap_decoder.generator.bias"].fill_(0.5)
    m.load_state_dict(sd)
    assert ps.intact() and float(ps.flat[ps.offsets["cap_decoder.generator.bias"]]) == 0.5
    # optimizer.zero_grad(set_to_none) then install_grads re-attaches the flat views
    for p in m.parameters():
        p.grad = None
    ps.install_grads()
    for n, p in m.named_parameters():
        if p.requires_grad:
            assert p.grad.data_ptr() == ps.gflat.data_ptr() + 4 * ps.offsets[n]


def test_preprocessor_pads_and_masks():
    from vct_amd.model.CapPreprocessor import CapPreprocessor
    pre = CapPreprocessor("ids", device=torch.device("cpu"))
    assert (pre.pad_id, pre.start_id, pre.end_id) == (0, 101, 102) and pre.tokenizer.vocab_size == 30522
    ids, mask = pre([[101, 7, 8, 102], [101, 9, 102]])
    assert ids.tolist() == [[101, 7, 8, 102], [101, 9, 102, 0]]
    assert mask.tolist() == [[False] * 4, [False, False, False, True]]
    ids2, _ = pre(torch.tensor([[101, 5, 102]]))
    assert ids2.dtype == torch.long
    try:
        pre(["a raw string needs the tokenizer files"]); assert False
    except RuntimeError:
        pass
    assert pre.tokenizer_error is None                 # "ids" asks for the pass-through: nothing failed
    # a named HF tokenizer whose files are not on this machine: the fallback is announced at construction and the reason
    # travels into the later encode() error
    import pytest
    with pytest.warns(RuntimeWarning, match="could not be loaded"):
        pre2 = CapPreprocessor("no-such-tokenizer-dir/bert-base-uncased", device=torch.device("cpu"))
    assert pre2.tokenizer_error
    with pytest.raises(RuntimeError, match="loading the tokenizer failed"):
        pre2(["a string"])
    assert pre2([[101, 3, 102]])[0].tolist() == [[101, 3, 102]]


def test_mask_builder_and_config(tmp_path):
    from vct_amd.utils import Config, generate_square_subsequent_mask
    assert np.array_equal(generate_square_subsequent_mask(19).numpy(), O.generate_square_subsequent_mask(19))
    p = tmp_path / "cfg.json"
    p.write_text(json.dumps({"model": SHIPPED_LIKE, "train": {"task": "caption"}}))
    cfg = Config(str(p))
    cfg.check()
    assert cfg.data["model"]["embed_dim"] == 768


def test_torch_like_initialisation_statistics():
    m = build_model(dict(SHIPPED_LIKE, embed_dim=256), 1000, "cpu", torch.float32)
    sd = m.state_dict()
    w = sd["cap_decoder.decoder.layers.0.linear1.weight"]       # kaiming_uniform(a=sqrt5): U(+-1/sqrt(in))
    assert abs(float(w.abs().max()) - 1 / np.sqrt(256)) < 2e-3
    a = sd["cap_decoder.decoder.layers.0.self_attn.in_proj_weight"]  # xavier_uniform on [3d, d]
    assert abs(float(a.abs().max()) - np.sqrt(6.0 / (4 * 256))) < 2e-3
    assert float(sd["cap_decoder.decoder.layers.0.self_attn.in_proj_bias"].abs().sum()) == 0.0
    assert abs(float(sd["cap_decoder.tgt_to_emb.weight"][1:].std()) - 1.0) < 0.02
    assert torch.equal(sd["cap_decoder.decoder.layers.0.linear1.weight"], sd["cap_decoder.decoder.layers.2.linear1.weight"])


def test_optimizer_ranges_left_by_the_weight_gradient_epilogues():
    """Host logic of the single-GPU schedule (trainer.FusedAdam): what the weight-gradient GEMMs' optimizer epilogues step is cut
    out of a flat range, the rest is split at the shadow-less token-embedding table, and the multi-range launch's table numbers its
    4096-element workgroups range after range (include/vct_hip.h, vct_adam_range)."""
    import ctypes
    from vct_amd import _lib, ops
    from vct_amd.trainer import FusedAdam
    opt = FusedAdam.__new__(FusedAdam)
    opt.skip = (10000, 20000)                                   # the embedding table: no bf16 shadow
    opt._dw_ranges = [(4096, 8192), (0, 1024), (30000, 40960), (4096, 8192)]     # registered twice: harmless
    left = opt._left_ranges(0, 50000)
    assert left == [(1024, 4096, True), (8192, 10000, True), (10000, 20000, False), (20000, 30000, True), (40960, 50000, True)]
    assert opt._left_ranges(4096, 8192) == [] and opt._left_ranges(9000, 21000) == [(9000, 10000, True), (10000, 20000, False), (20000, 21000, True)]
    table, n, blocks = ops.adam_ranges_table(left, "cpu")
    assert n == 5 and table.numel() == n * ctypes.sizeof(_lib.AdamRange)
    arr = (_lib.AdamRange * n).from_buffer_copy(bytes(table.numpy().tobytes()))
    want_blk, at = [], 0
    for a, b, _sh in left:
        want_blk.append(at); at += (b - a + 4095) // 4096
    assert [r.blk0 for r in arr] == want_blk and blocks == at
    assert [(r.begin, r.end, r.shadow) for r in arr] == [(a, b, int(sh)) for a, b, sh in left]
