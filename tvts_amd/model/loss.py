"""Drop-in for v2/model/loss.py: NormSoftmaxLoss(temperature=0.05) on a similarity matrix, HIP kernels."""
import torch
import torch.nn as nn

from .. import hip as K


class _InfoNCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, temperature):
        G = x.shape[0]
        if x.shape[0] != x.shape[1]:
            raise ValueError("NormSoftmaxLoss expects a square similarity matrix")
        xs = (x.contiguous().float() / temperature)
        lse = torch.empty(2 * G, dtype=torch.float32, device=x.device)
        dx = torch.empty(G, G, dtype=torch.float32, device=x.device)
        loss = torch.zeros(1, dtype=torch.float32, device=x.device)
        K.infonce(xs, lse, dx, loss)
        ctx.save_for_backward(dx)
        ctx.temperature = temperature
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * (g / ctx.temperature), None


class NormSoftmaxLoss(nn.Module):
    def __init__(self, temperature=0.05):
        super().__init__()
        self.temperature = temperature

    def forward(self, x):
        """x: similarity matrix N x N in [-1, 1] (rows = videos); both softmax directions, no 1/2 (loss.py:13-25)."""
        return _InfoNCEFn.apply(x, self.temperature)
