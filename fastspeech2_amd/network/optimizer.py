"""Noam learning-rate schedule around a torch optimiser — the counterpart of network/optimizer.py:5-59
(`NoamOpt`, `get_std_opt`) and :62-85 (`ScheduledOptimizer`), same constructor arguments, methods and
state_dict keys so snapshots interchange.  The parameter update itself is torch.optim.Adam on the
device (not a custom kernel)."""
import torch


class NoamOpt(object):
    """lr(step) = factor * model_size^-0.5 * min(step^-0.5, step * warmup^-1.5)"""

    _SCALARS = ("_step", "warmup", "factor", "model_size", "_rate")      # snapshot keys besides "optimizer"

    def __init__(self, model_size, factor, warmup, optimizer: torch.optim.Optimizer):
        self._step, self._rate = 0, 0
        self.model_size, self.factor, self.warmup = model_size, factor, warmup
        self.optimizer = optimizer

    param_group = property(lambda self: self.optimizer.param_groups)

    def rate(self, step=None):
        """learning rate at `step` (default: the current step count)"""
        s = self._step if step is None else step
        warm, decay = s * self.warmup ** (-1.5), s ** (-0.5)
        return self.factor * self.model_size ** (-0.5) * min(decay, warm)

    def step(self):
        """advance the schedule, write the rate into every param group, then step the wrapped optimiser"""
        self._step += 1
        lr = self._rate = self.rate()
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        self.optimizer.step()

    def zero_grad(self):
        self.optimizer.zero_grad()

    def state_dict(self):
        sd = {k: getattr(self, k) for k in self._SCALARS}
        sd["optimizer"] = self.optimizer.state_dict()
        return sd

    def load_state_dict(self, state_dict: dict):
        for key, value in state_dict.items():
            if key == "optimizer":
                self.optimizer.load_state_dict(value)
            else:
                setattr(self, key, value)


def get_std_opt(model: torch.nn.Module, d_model, warmup, factor):
    """Adam(betas 0.9/0.98, eps 1e-9) under the Noam schedule (network/optimizer.py:57-59)."""
    params = list(model.parameters())
    fused = bool(params) and all(p.is_cuda for p in params)      # one multi-tensor kernel on the device, same update rule
    return NoamOpt(d_model, factor, warmup, torch.optim.Adam(params, lr=0, betas=(0.9, 0.98), eps=1e-9, fused=fused))


class ScheduledOptimizer:
    """The second scheduler of the reference (network/optimizer.py:62-85): lr = d^-0.5 * min(s^-0.5, s * w^-1.5)."""

    def __init__(self, optimizer, decoder_hidden_dim, warm_up_steps, current_step):
        self.optimizer = optimizer
        self.decoder_hidden_dim = decoder_hidden_dim
        self.warm_up_steps = warm_up_steps
        self.current_steps = current_step
        self.init_lr = decoder_hidden_dim ** -0.5

    def update_lr(self):
        self.current_steps += 1
        s = self.current_steps
        lr = self.init_lr * min(s ** -0.5, self.warm_up_steps ** -1.5 * s)
        for group in self.optimizer.param_groups:
            group["lr"] = lr

    def step_update_lr(self):
        self.update_lr()
        self.optimizer.step()

    def zero_grad(self):
        self.optimizer.zero_grad()
