"""Element-wise forms of the three losses the shipped configs name (registry API of det3d/models/losses/losses.py:
`SigmoidFocalLoss` :269-336, `WeightedSmoothL1Loss` :146-211, `WeightedSoftmaxClassificationLoss` :407-446 -- same
class names and constructor keywords, so config dicts build unchanged).

On the GPU the training step never comes here: `d3d_head_loss` (csrc/d3d_loss.hip) evaluates a whole task group,
forward and gradient, in two launches.  These modules are what the CPU plumbing configuration (BASELINE configs[0])
and the unfused path of `MultiGroupHead.loss` call.  The arithmetic is written from the definitions:

  focal      FL(x, t) = a_t * (1 - p_t)^gamma * BCE(x, t),   BCE through softplus so that large |x| is safe
  smooth-L1  Huber with threshold 1 / sigma^2, per box-code element; the reference reads a `code_weights` keyword and
             then never applies it (losses.py:173) -- kept, parity means replicating that
  softmax    cross entropy of the direction bins against the arg-max of a one-hot target
"""
import torch
from torch import nn
from torch.nn import functional as F

from .registry import LOSSES


def focal_terms(logits, onehot, alpha, gamma):
    """per-element sigmoid focal loss; `onehot` has the logits' shape and dtype-compatible 0/1 entries"""
    t = onehot.to(logits.dtype)
    bce = F.softplus(logits) - logits * t            # = -[t log p + (1 - t) log(1 - p)]
    p = torch.sigmoid(logits)
    miss = torch.where(t > 0, 1.0 - p, p)            # 1 - p_t
    out = bce if not gamma else bce * miss.pow(gamma)
    if alpha is not None:
        out = out * torch.where(t > 0, alpha, 1.0 - alpha).to(out.dtype)
    return out


def huber_terms(pred, target, sigma):
    """0.5 (sigma d)^2 below |d| = 1 / sigma^2, |d| - 0.5 / sigma^2 above"""
    d = (pred - target).abs()
    knee = 1.0 / (sigma * sigma)
    return torch.where(d <= knee, 0.5 * (sigma * d) ** 2, d - 0.5 * knee)


@LOSSES.register_module
class WeightedSmoothL1Loss(nn.Module):
    def __init__(self, sigma=3.0, reduction="mean", code_weights=None, codewise=True, loss_weight=1.0):
        super().__init__()
        self._sigma, self._codewise = sigma, codewise
        self._reduction, self._loss_weight = reduction, loss_weight
        self._code_weights = None  # accepted and dropped, like the reference

    def forward(self, prediction_tensor, target_tensor, weights=None):
        terms = huber_terms(prediction_tensor, target_tensor, self._sigma)
        if not self._codewise:
            terms = terms.sum(2)
            return terms if weights is None else terms * weights
        return terms if weights is None else terms * weights.unsqueeze(-1)


@LOSSES.register_module
class SigmoidFocalLoss(nn.Module):
    def __init__(self, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0):
        super().__init__()
        self._alpha, self._gamma = alpha, gamma
        self._reduction, self._loss_weight = reduction, loss_weight

    def forward(self, prediction_tensor, target_tensor, weights=None, class_indices=None):
        return focal_terms(prediction_tensor, target_tensor, self._alpha, self._gamma) * weights.unsqueeze(2)


@LOSSES.register_module
class WeightedSoftmaxClassificationLoss(nn.Module):
    def __init__(self, logit_scale=1.0, loss_weight=1.0, name=""):
        super().__init__()
        self.name = name
        self._loss_weight, self._logit_scale = loss_weight, logit_scale

    def forward(self, prediction_tensor, target_tensor, weights):
        bins = prediction_tensor.shape[-1]
        logits = (prediction_tensor / self._logit_scale).reshape(-1, bins)
        ce = F.cross_entropy(logits, target_tensor.reshape(-1, bins).argmax(1), reduction="none")
        return ce.reshape(weights.shape) * weights
