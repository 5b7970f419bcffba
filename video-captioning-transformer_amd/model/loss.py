"""SCELoss -- reference: model/loss.py:69-92.  Callable on (logits[N,V], labels[N]) like the
reference's module; the arithmetic (loss AND d loss / d logits) is the vct_sce_loss kernel."""
import torch
import torch.nn as nn

from .. import ops


class _SCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, alpha, pad):
        N, V = logits.shape
        Vp = (V + 31) // 32 * 32
        lg = torch.zeros(N, Vp, dtype=logits.dtype, device=logits.device)
        lg[:, :V] = logits
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        dl = torch.empty_like(lg)
        ws = torch.empty(2 * N + 2, dtype=torch.float32, device=logits.device)
        ops.sce_loss(lg, V, labels.view(N, 1), 1, pad, alpha, loss, dl, ws)
        ctx.save_for_backward(dl)
        ctx.V = V
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl[:, :ctx.V] * g, None, None, None


class SCELoss(nn.Module):
    def __init__(self, alpha, beta, ignore_index, num_classes=10, device=None):
        super().__init__()
        assert abs(alpha + beta - 1.0) < 1e-6, "the kernel implements beta = 1 - alpha (CapDecoder.py:31)"
        self.alpha, self.beta, self.ignore_index, self.num_classes = alpha, beta, ignore_index, num_classes

    def forward(self, pred: torch.Tensor, labels: torch.Tensor):
        return _SCEFn.apply(pred.contiguous(), labels.contiguous(), float(self.alpha), int(self.ignore_index))
