"""The three losses the shipped configs use (det3d/models/losses/losses.py):
  SigmoidFocalLoss                   :269-336
  WeightedSmoothL1Loss               :146-211  (code_weights ignored, as in the reference :173)
  WeightedSoftmaxClassificationLoss  :407-446
Plain PyTorch elementwise math (fused well enough by the framework; not a kernel target)."""
import torch
from torch import nn

from .registry import LOSSES


def _sigmoid_cross_entropy_with_logits(logits, labels):
    loss = torch.clamp(logits, min=0) - logits * labels.type_as(logits)
    loss = loss + torch.log1p(torch.exp(-torch.abs(logits)))
    return loss


def _softmax_cross_entropy_with_logits(logits, labels):
    dims = list(range(logits.dim()))
    logits = logits.permute(*([0] + [dims[-1]] + dims[1:-1]))
    return nn.functional.cross_entropy(logits, labels.max(dim=-1)[1], reduction="none")


@LOSSES.register_module
class WeightedSmoothL1Loss(nn.Module):
    def __init__(self, sigma=3.0, reduction="mean", code_weights=None, codewise=True, loss_weight=1.0):
        super().__init__()
        self._sigma = sigma
        self._code_weights = None  # the reference discards code_weights
        self._codewise = codewise
        self._reduction = reduction
        self._loss_weight = loss_weight

    def forward(self, prediction_tensor, target_tensor, weights=None):
        diff = prediction_tensor - target_tensor
        abs_diff = torch.abs(diff)
        lt = torch.le(abs_diff, 1 / (self._sigma ** 2)).type_as(abs_diff)
        loss = lt * 0.5 * torch.pow(abs_diff * self._sigma, 2) + \
            (abs_diff - 0.5 / (self._sigma ** 2)) * (1.0 - lt)
        if self._codewise:
            out = loss
            if weights is not None:
                out = out * weights.unsqueeze(-1)
        else:
            out = torch.sum(loss, 2)
            if weights is not None:
                out = out * weights
        return out


@LOSSES.register_module
class SigmoidFocalLoss(nn.Module):
    def __init__(self, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0):
        super().__init__()
        self._alpha = alpha
        self._gamma = gamma
        self._reduction = reduction
        self._loss_weight = loss_weight

    def forward(self, prediction_tensor, target_tensor, weights=None, class_indices=None):
        weights = weights.unsqueeze(2)
        ce = _sigmoid_cross_entropy_with_logits(labels=target_tensor, logits=prediction_tensor)
        p = torch.sigmoid(prediction_tensor)
        p_t = target_tensor * p + (1 - target_tensor) * (1 - p)
        mod = torch.pow(1.0 - p_t, self._gamma) if self._gamma else 1.0
        alpha = 1.0
        if self._alpha is not None:
            alpha = target_tensor * self._alpha + (1 - target_tensor) * (1 - self._alpha)
        return mod * alpha * ce * weights


@LOSSES.register_module
class WeightedSoftmaxClassificationLoss(nn.Module):
    def __init__(self, logit_scale=1.0, loss_weight=1.0, name=""):
        super().__init__()
        self.name = name
        self._loss_weight = loss_weight
        self._logit_scale = logit_scale

    def forward(self, prediction_tensor, target_tensor, weights):
        num_classes = prediction_tensor.shape[-1]
        prediction_tensor = torch.div(prediction_tensor, self._logit_scale)
        ce = _softmax_cross_entropy_with_logits(labels=target_tensor.view(-1, num_classes),
                                                logits=prediction_tensor.view(-1, num_classes))
        return ce.view(weights.shape) * weights
