"""Pin the CPU oracle (oracle/vct_oracle.py) against the golden vectors produced by the real
reference (oracle/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

import vct_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _params(z, cfg):
    p = {k[len("param/"):]: z[k] for k in z.files if k.startswith("param/")}
    p[O.ENC + "temp_emb.pe"] = O.encoder_pos_table(512, cfg["d"])
    p[O.DEC + "positional_encoding.pos_embedding"] = O.decoder_pos_table(5000, cfg["d"])
    return p


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def test_tables_and_masks(golden_dir):
    z = _load(golden_dir, "tiny_train.npz")
    tab = O.decoder_pos_table(5000, 64)
    np.testing.assert_allclose(tab[:64], z["pos_embedding_head"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(tab[4990:], z["pos_embedding_tail"], rtol=0, atol=5e-4)  # sin/cos of ~5000 rad in fp32
    np.testing.assert_allclose(O.encoder_pos_table(512, 64), z["temp_pe"], rtol=0, atol=2e-5)
    m = _load(golden_dir, "mha.npz")
    assert np.array_equal(O.generate_square_subsequent_mask(6), m["mask6"])
    assert np.array_equal(O.generate_square_subsequent_mask(19), m["mask19"])


def test_mha_merged_masks(golden_dir):
    z = _load(golden_dir, "mha.npz")
    add = O.generate_square_subsequent_mask(6)[None, None] + O._bool_to_add(z["kpm"], np.float32)[:, None, None, :]
    y, _ = O.mha_forward(z["xq"], z["xq"], z["w_in"], z["b_in"], z["w_o"], z["b_o"], 4, add)
    assert relerr(y, z["y_self"]) < 2e-6
    y, _ = O.mha_forward(z["xq"], z["xm"], z["w_in"], z["b_in"], z["w_o"], z["b_o"], 4, None)
    assert relerr(y, z["y_cross"]) < 2e-6


@pytest.mark.parametrize("alpha", [0.5, 0.3, 1.0])
def test_sce_loss_and_grad(golden_dir, alpha):
    z = _load(golden_dir, "sce_loss.npz")
    loss, dl = O.sce_loss(z["logits"], z["labels"], alpha, 0)
    assert abs(loss - z[f"a{alpha}/loss"]) < 2e-6 * abs(z[f"a{alpha}/loss"])
    assert relerr(dl, z[f"a{alpha}/dlogits"]) < 5e-6
    # fp64 oracle agrees too (noise floor of the fp32 reference)
    loss64, _ = O.sce_loss(z["logits"].astype(np.float64), z["labels"], alpha, 0)
    assert abs(loss64 - z[f"a{alpha}/loss"]) < 5e-6 * abs(loss64)


def test_tiny_model_forward_backward_adam(golden_dir):
    z = _load(golden_dir, "tiny_train.npz")
    mc = json.loads(str(z["model_config"]))
    cfg = O.cfg_from_model_config(mc, int(z["vocab"]))
    p = _params(z, cfg)
    # state-dict surface: every trainable key/shape of the reference exists in the oracle's params
    keys = json.loads(str(z["state_keys"]))
    for k, shp in keys.items():
        assert k in p and list(p[k].shape) == shp, k
    feats, mask, ids = z["feats"], z["mask"], z["ids"]
    mem, kpm, c_enc, enc_layers = O.mm_encoder_forward(p, cfg, feats, mask, return_cache=True, return_layers=True)
    assert relerr(enc_layers[0], z["act/mm_src"]) < 1e-6
    for l in range(cfg["enc_layers"]):
        assert relerr(enc_layers[l + 1], z[f"act/enc_layer{l}"]) < 5e-6
    assert relerr(mem, z["act/memory"]) < 5e-6
    logits, loss, c_dec, dec_layers = O.cap_decoder_forward(p, cfg, mem, ids, return_cache=True, return_layers=True)
    assert relerr(dec_layers[0], z["act/tgt_emb"]) < 1e-6
    for l in range(cfg["dec_layers"]):
        assert relerr(dec_layers[l + 1], z[f"act/dec_layer{l}"]) < 5e-6
    assert relerr(dec_layers[-1], z["act/dec_out"]) < 5e-6
    assert relerr(logits, z["act/logits"]) < 5e-6
    assert abs(loss - float(z["loss"])) < 5e-6 * float(z["loss"])
    loss2, grads, _ = O.caption_loss_and_grads(p, cfg, feats, mask, ids)
    gk = [k[len("grad/"):] for k in z.files if k.startswith("grad/")]
    assert sorted(gk) == sorted(grads)
    for k in gk:
        assert relerr(grads[k], z["grad/" + k]) < 3e-5, k
    assert np.all(grads[O.DEC + "tgt_to_emb.weight"][0] == 0)
    newp = O.adam_step(p, grads, {}, lr=1e-4)
    for k in gk:
        # after one Adam step every element moved by ~lr*sign(g); compare the update, not just the value
        upd_ref = z["adam1/" + k].astype(np.float64) - p[k]
        upd = newp[k].astype(np.float64) - p[k]
        big = np.abs(z["grad/" + k]) > 1e-6   # elements with |g| ~ eps are sign-unstable in fp32
        assert np.abs(upd - upd_ref)[big].max(initial=0) < 2e-6, k


def test_tiny_model_ce_relu_variant(golden_dir):
    z = _load(golden_dir, "tiny_train_ce_relu.npz")
    mc = json.loads(str(z["model_config"]))
    cfg = O.cfg_from_model_config(mc, int(z["vocab"]))
    p = O.init_params(cfg, seed=12)
    loss, grads, logits = O.caption_loss_and_grads(p, cfg, z["feats"], z["mask"], z["ids"])
    assert relerr(logits, z["logits"]) < 5e-6
    assert abs(loss - float(z["loss"])) < 5e-6 * float(z["loss"])
    for k in [k[len("grad/"):] for k in z.files if k.startswith("grad/")]:
        assert relerr(grads[k], z["grad/" + k]) < 3e-5, k


def test_greedy_decode_ids_and_logits(golden_dir):
    z = _load(golden_dir, "tiny_decode.npz")
    mc = json.loads(str(z["model_config"]))
    cfg = O.cfg_from_model_config(mc, int(z["vocab"]))
    p = O.init_params(cfg, seed=int(z["param_seed"]))
    for tag in ("b1", "b3"):
        feats = z[f"{tag}/feats"]
        mem = O.mm_encoder_forward(p, cfg, feats, None)[0]
        assert relerr(mem, z[f"{tag}/memory_eval"]) < 1e-5   # eval fast path vs train path: fp32 noise only
        ys_ref = z[f"{tag}/ys"]
        ys = O.greedy_decode_ids(p, cfg, feats, None, max_len=12)
        assert np.array_equal(ys, ys_ref)
        ys2 = O.greedy_decode_ids(p, cfg, feats, np.zeros(feats.shape[:2], bool), max_len=12)
        assert np.array_equal(ys2, ys_ref)
        ls = z[f"{tag}/logits_steps"]
        for t in range(ls.shape[1]):   # teacher-forced per-step logits
            lg = O.decode_word(p, cfg, mem, ys_ref[:, :t + 1])
            assert relerr(lg, ls[:, t]) < 1e-5
        caps = json.loads(str(z[f"{tag}/captions"]))
        mine = [" ".join(str(i) for i in row) for row in O.ids_to_caption_ids(ys)]
        assert mine == caps


def test_eval_fast_path_with_padding_quirk(golden_dir):
    """Appendix C.6: in eval()/no_grad with a real padding mask torch's nested-tensor fast path yields
    different values at PADDED positions only; un-padded positions match the train-path oracle."""
    z = _load(golden_dir, "tiny_decode.npz")
    mc = json.loads(str(z["model_config"]))
    cfg = O.cfg_from_model_config(mc, int(z["vocab"]))
    p = O.init_params(cfg, seed=int(z["param_seed"]))
    feats, mask = z["masked/feats"], z["masked/mask"]
    mem, kpm = O.mm_encoder_forward(p, cfg, feats, mask)
    ref = z["masked/memory_eval"]
    assert relerr(mem[~kpm], ref[~kpm]) < 1e-5


def test_cfgA_full_size_slices(golden_dir):
    """cfg-A (d=512, 2 enc + 2 dec, V=30522, B=8, T=12, S=20): BASELINE.json configs[0]."""
    z = _load(golden_dir, "cfgA_slices.npz")
    mc = json.loads(str(z["model_config"]))
    cfg = O.cfg_from_model_config(mc, int(z["vocab"]))
    p = O.init_params(cfg, seed=int(z["param_seed"]))
    n_train = sum(v.size for k, v in p.items() if k not in O.BUFFER_KEYS)
    assert n_train == int(z["n_params"]) == 46262586
    feats, mask, ids = O.synthetic_batch(8, 12, 512, 20, int(z["vocab"]), seed=int(z["batch_seed"]))
    loss, grads, logits = O.caption_loss_and_grads(p, cfg, feats, mask, ids)
    assert abs(loss - float(z["loss"])) < 1e-5 * float(z["loss"])
    assert relerr(logits[:, :, :96], z["logits_head"]) < 1e-5
    assert np.array_equal(logits.argmax(-1), z["logits_argmax"])
    names = json.loads(str(z["grad_names"]))
    for i, k in enumerate(names):
        n = np.linalg.norm(grads[k].astype(np.float64))
        assert abs(n - z["grad_norms"][i]) < 1e-4 * z["grad_norms"][i] + 1e-9, k
        np.testing.assert_allclose(np.resize(grads[k].reshape(-1)[:32], 32), z["grad_heads"][i],
                                   rtol=2e-3, atol=1e-7 + 1e-4 * np.abs(z["grad_heads"][i]).max())


def test_cfgA_greedy_decode(golden_dir):
    z = _load(golden_dir, "cfgA_decode.npz")
    cfg = O.cfg_from_model_config(json.loads(str(_load(golden_dir, "cfgA_slices.npz")["model_config"])), 30522)
    p = O.init_params(cfg, seed=int(z["param_seed"]))
    feats = O.synthetic_batch(4, 12, 512, 20, 30522, seed=int(z["feats_seed"]))[0]
    ys = O.greedy_decode_ids(p, cfg, feats, None, max_len=30)
    ref = z["ys"]
    # free-running ids must match wherever the reference's own top-2 margin is above fp32 noise
    assert ys.shape[1] <= ref.shape[1]
    assert float(z["margins"].min()) > 1e-4
    assert np.array_equal(ys, ref[:, :ys.shape[1]])


def test_torch_cpu_baseline_model_matches_the_oracle():
    """oracle/torch_ref.py (the module graph bench.py times as the CPU baseline) == the numpy oracle on a ragged batch:
    loss, logits and every parameter gradient (dropout 0)."""
    import torch
    import torch_ref as TR
    mc = {"modal_shape": [24], "embed_dim": 32, "activation": "gelu",
          "video_encoder": {"layer": 2, "nhead": 4, "feedforward": 48},
          "caption_decoder": {"layer": 2, "nhead": 4, "feedforward": 48, "sce_loss_alpha": 0.5}}
    cfg = O.cfg_from_model_config(mc, 131)
    p = O.init_params(cfg, seed=5)
    feats, mask, ids = O.synthetic_batch(5, 6, 24, 8, 131, seed=2, ragged=True)
    ref_loss, ref_grads, ref_logits = O.caption_loss_and_grads(p, cfg, feats, mask, ids)
    torch.manual_seed(0)
    m = TR.RefCaptionModel(cfg, dropout=0.0)
    m.load_oracle_params(p)
    m.train()
    logits, loss = m(torch.from_numpy(feats), torch.from_numpy(mask), torch.from_numpy(ids))
    loss.backward()
    assert abs(float(loss) - ref_loss) < 1e-5 * abs(ref_loss)
    np.testing.assert_allclose(logits.detach().numpy().reshape(ref_logits.shape), ref_logits, rtol=2e-4, atol=2e-5)
    named = dict(m.named_parameters())
    for k, g in ref_grads.items():
        mine = named[k].grad.numpy()
        err = np.linalg.norm(mine - g) / max(np.linalg.norm(g), 1e-30)
        assert err < 2e-4, (k, err)


def test_oracle_vs_reference_cfgD_and_cfgB_decode():
    """The deep configuration (6 + 6 layers, d = 1024, head_dim 128, 32 frames, 40 tokens, V = 30522, ragged batch of 8) and the
    d=512 greedy decode at batch 1 / 16, recorded from the real reference by oracle/make_golden_cfgD.py."""
    import json
    GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(GOLDEN, "cfgD_slices.npz"), allow_pickle=False)
    mc, V = json.loads(str(z["model_config"])), int(z["vocab"])
    cfg = O.cfg_from_model_config(mc, V)
    p = O.init_params(cfg, seed=int(z["param_seed"]))
    f, mk, ids = O.synthetic_batch(8, 32, 512, 40, V, seed=int(z["batch_seed"]), ragged=True)
    loss, grads, logits = O.caption_loss_and_grads(p, cfg, f, mk, ids)
    assert abs(loss - float(z["loss"])) < 2e-6 * float(z["loss"])
    lg = logits.reshape(8, 39, V)
    np.testing.assert_allclose(lg[:, :, :64], z["logits_head"], rtol=2e-4, atol=2e-5)
    names = json.loads(str(z["grad_names"]))
    for i, k in enumerate(names):
        n = float(np.linalg.norm(grads[k].astype(np.float64)))
        assert abs(n - z["grad_norms"][i]) < 3e-5 * z["grad_norms"][i] + 1e-12, (k, n, z["grad_norms"][i])
    zd = np.load(os.path.join(GOLDEN, "cfgB_decode.npz"), allow_pickle=False)
    mcB = json.loads(str(zd["model_config"]))
    cfgB = O.cfg_from_model_config(mcB, V)
    pB = O.init_params(cfgB, seed=int(zd["param_seed"]))
    fB = O.synthetic_batch(1, 12, 512, 20, V, seed=int(zd["feats_seed_b1"]))[0]
    ys = O.greedy_decode_ids(pB, cfgB, fB, None, max_len=30)
    m1 = zd["margins_b1"][0]
    low = np.nonzero(m1 < 5e-5)[0]
    upto = int(low[0]) + 1 if low.size else ys.shape[1]
    assert np.array_equal(ys[0, :upto], zd["ys_b1"][0, :upto])
