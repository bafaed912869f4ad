"""KL-weight and learning-rate schedules of the reference trainer (polygon/vae/vae_misc.py:6-61)."""
import math

from torch.optim.lr_scheduler import _LRScheduler


class KLAnnealer:
    """Linear KL weight: w_start until epoch kl_start, then a constant increment per epoch reaching w_end at n_epoch
    (vae_misc.py:6-22; divides by n_epoch - kl_start, so equal values raise ZeroDivisionError as the reference does)."""

    def __init__(self, n_epoch, kl_start=0, kl_w_start=0, kl_w_end=1):
        self.i_start = kl_start
        self.w_start = kl_w_start
        self.w_max = kl_w_end
        self.n_epoch = n_epoch
        self.inc = (self.w_max - self.w_start) / (self.n_epoch - self.i_start)

    def __call__(self, i):
        steps = max(i - self.i_start, 0)
        return self.w_start + steps * self.inc


class CosineAnnealingLRWithRestart(_LRScheduler):
    """SGDR (vae_misc.py:25-61): lr = lr_end + (base - lr_end)(1 + cos(pi cur / t_end)) / 2 with cur advanced by every
    step() (the constructor performs one), restarting when cur reaches t_end, which is then scaled by lr_n_mult."""

    def __init__(self, optimizer, lr_n_period=10, lr_n_restarts=6, lr_n_mult=1, lr_end=0.0003):
        self.n_period = lr_n_period
        self.n_mult = lr_n_mult
        self.lr_end = lr_end
        self.current_epoch = 0
        self.t_end = self.n_period
        super().__init__(optimizer, -1)

    def get_lr(self):
        scale = (1 + math.cos(math.pi * self.current_epoch / self.t_end)) / 2
        return [self.lr_end + (base - self.lr_end) * scale for base in self.base_lrs]

    def step(self, epoch=None):
        self.last_epoch = self.last_epoch + 1 if epoch is None else epoch
        self.current_epoch += 1
        for group, lr in zip(self.optimizer.param_groups, self.get_lr()):
            group['lr'] = lr
        if self.current_epoch == self.t_end:
            self.current_epoch = 0
            self.t_end = self.n_mult * self.t_end
