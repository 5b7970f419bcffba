"""Caption -> padded id tensor + pad mask (reference: model/CapPreprocessor.py:7-36).  Host-side
input preparation, not accelerated.  Accepts raw strings (needs the HF tokenizer files locally) or
pre-tokenised id sequences (lists / tensors), which is what the synthetic benchmarks feed."""
import warnings
from typing import List, Sequence, Tuple, Union

import torch


class _IdTokenizer:
    """Stand-in when the `bert-base-uncased` vocabulary is not available offline: ids pass through."""
    vocab_size = 30522
    reason = None          # why the real tokenizer is absent (set by CapPreprocessor)

    def convert_tokens_to_ids(self, tok):
        return {"[PAD]": 0, "[CLS]": 101, "[SEP]": 102}[tok]

    def convert_ids_to_tokens(self, ids):
        return [str(i) for i in ids]

    def convert_tokens_to_string(self, toks):
        return " ".join(toks)

    def encode(self, text, **_):
        raise RuntimeError("no tokenizer vocabulary available offline: pass pre-tokenised id lists instead of strings"
                           + (f" (loading the tokenizer failed with: {self.reason})" if self.reason else ""))


class CapPreprocessor:
    def __init__(self, tokenizer_type, device=torch.device("cuda"), vocab_size=None):
        self.tokenizer_type, self.device = tokenizer_type, device
        self.tokenizer_error = None         # set when the requested HF tokenizer could not be loaded (id pass-through in use)
        tok = None
        if isinstance(tokenizer_type, str) and tokenizer_type not in ("stub", "ids"):
            try:
                from transformers import AutoTokenizer
                tok = AutoTokenizer.from_pretrained(tokenizer_type, local_files_only=True)
            except Exception as e:      # no vocabulary files on this machine: say so NOW, not at the first encode()
                tok = None
                self.tokenizer_error = f"{type(e).__name__}: {e}"
                warnings.warn(f"CapPreprocessor: tokenizer '{tokenizer_type}' could not be loaded from local files "
                              f"({self.tokenizer_error}); captions must be passed as id sequences (strings will raise)",
                              RuntimeWarning, stacklevel=2)
        if tok is None:
            tok = _IdTokenizer()
            tok.reason = getattr(self, "tokenizer_error", None)
            if vocab_size is not None:
                tok.vocab_size = vocab_size
        self.tokenizer = tok
        self.pad_id = tok.convert_tokens_to_ids("[PAD]")
        self.start_id = tok.convert_tokens_to_ids("[CLS]")
        self.end_id = tok.convert_tokens_to_ids("[SEP]")

    def __call__(self, captions: Union[torch.Tensor, Sequence]) -> Tuple[torch.Tensor, torch.Tensor]:
        if isinstance(captions, torch.Tensor):
            text_ts = captions.to(self.device, dtype=torch.long)
        else:
            rows: List[List[int]] = []
            for c in captions:
                rows.append(self.tokenizer.encode(c) if isinstance(c, str) else [int(t) for t in c])
            max_len = max(len(r) for r in rows)
            host = torch.full((len(rows), max_len), self.pad_id, dtype=torch.long)
            for i, r in enumerate(rows):
                host[i, :len(r)] = torch.tensor(r, dtype=torch.long)
            text_ts = host.to(self.device)   # ONE H2D copy (the reference does B of them, CapPreprocessor.py:28,34)
        return text_ts, text_ts == self.pad_id
