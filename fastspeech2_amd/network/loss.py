"""Duration loss in the log domain (network/loss.py:5-14)."""
import torch
import torch.nn as nn


class DurationPredictorLoss(nn.Module):
    def __init__(self, offset: float = 1.0):
        super().__init__()
        self.offset = offset
        self.criterion = nn.MSELoss()

    def forward(self, outputs: torch.Tensor, targets: torch.Tensor):
        return self.criterion(outputs, torch.log(targets.float() + self.offset))
