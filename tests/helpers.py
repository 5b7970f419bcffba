import json
import os

import numpy as np
import torch

import vct_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def model_config_of(z):
    return json.loads(str(z["model_config"]))


def build_model(mc, vocab, device, compute_dtype, params=None):
    """Our MMT4Caption with the given cfg['model'] block; optionally load oracle-style numpy params."""
    from vct_amd.model import MMT4Caption
    mc = dict(mc)
    mc["tokenizer"] = "ids"
    mc["vocab_size"] = vocab
    m = MMT4Caption(mc, device=torch.device(device), compute_dtype=compute_dtype)
    m.mode("caption")
    if params is not None:
        sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()}
        res = m.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys, res.unexpected_keys
        assert all(k.startswith("matching") for k in res.missing_keys), res.missing_keys
    return m


def golden_params(z, cfg):
    p = {k[len("param/"):]: z[k] for k in z.files if k.startswith("param/")}
    p[O.ENC + "temp_emb.pe"] = O.encoder_pos_table(512, cfg["d"])
    p[O.DEC + "positional_encoding.pos_embedding"] = O.decoder_pos_table(5000, cfg["d"])
    return p


def rel(a, b):
    a = np.asarray(a.detach().float().cpu() if torch.is_tensor(a) else a, np.float64)
    b = np.asarray(b.detach().float().cpu() if torch.is_tensor(b) else b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
