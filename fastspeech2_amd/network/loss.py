"""Duration loss of the variance adaptor: mean squared error between the predictor's output and
log(duration + offset), i.e. durations are regressed in the log domain (what network/loss.py:5-14
defines).  On the training path the gradient of this term is produced by `fs2_loss_grad` (mode 2);
this module yields the reported value and serves callers that build the loss themselves."""
import torch
import torch.nn.functional as F
from torch import nn


def log_domain_targets(durations: torch.Tensor, offset: float) -> torch.Tensor:
    """integer frame counts -> regression targets log(d + offset), fp32"""
    return (durations.to(torch.float32) + offset).log()


class DurationPredictorLoss(nn.Module):
    """`loss(outputs, targets)`: outputs are log-domain predictions (B, T) or packed (M,), targets integer
    durations of the same shape.  `offset` (default 1.0) keeps log finite for zero-length phones."""

    def __init__(self, offset: float = 1.0):
        nn.Module.__init__(self)
        self.offset = float(offset)

    def extra_repr(self) -> str:
        return "offset=%g" % self.offset

    def forward(self, outputs: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        return F.mse_loss(outputs, log_domain_targets(targets, self.offset))
