"""Optimizer + LR schedules of the pretrain step -- oracle (TEST INFRASTRUCTURE).

UPSTREAM, NOT IN MOUNT: ``AdamW``, ``WarmupLinearSchedule`` and
``WarmupConstantSchedule`` come from ``transformers.pytorch_transformers``
(call sites ``tasks/viewpoint_select/pretrain.py:26-30, 108-139``).  Restated
from the published pytorch-transformers 1.x rule, which differs from
``torch.optim.AdamW`` in two places: ``eps`` is added to the UN-corrected
``sqrt(v)`` (the bias corrections are folded into the step size), and the
decoupled weight decay ``p -= lr * wd * p`` is applied AFTER the Adam move
using the already-moved parameter.
"""
import math

import torch
from torch.optim import Optimizer
from torch.optim.lr_scheduler import LambdaLR


class AdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {}".format(lr))
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None if closure is None else closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                m, v = st["exp_avg"], st["exp_avg_sq"]
                m.mul_(b1).add_(g, alpha=1.0 - b1)
                v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
                denom = v.sqrt().add_(group["eps"])
                step_size = group["lr"]
                if group["correct_bias"]:
                    step_size = step_size * math.sqrt(1.0 - b2 ** st["step"]) / (1.0 - b1 ** st["step"])
                p.addcdiv_(m, denom, value=-step_size)
                if group["weight_decay"] > 0.0:
                    p.add_(p, alpha=-group["lr"] * group["weight_decay"])
        return loss


def warmup_linear_lambda(warmup_steps, t_total):
    def f(step):
        if step < warmup_steps:
            return float(step) / float(max(1, warmup_steps))
        return max(0.0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))

    return f


def warmup_constant_lambda(warmup_steps):
    def f(step):
        if step < warmup_steps:
            return float(step) / float(max(1.0, warmup_steps))
        return 1.0

    return f


class WarmupLinearSchedule(LambdaLR):
    def __init__(self, optimizer, warmup_steps, t_total, last_epoch=-1):
        super().__init__(optimizer, warmup_linear_lambda(warmup_steps, t_total), last_epoch=last_epoch)


class WarmupConstantSchedule(LambdaLR):
    def __init__(self, optimizer, warmup_steps, last_epoch=-1):
        super().__init__(optimizer, warmup_constant_lambda(warmup_steps), last_epoch=last_epoch)


def grouped_parameters(model, weight_decay):
    """The no-decay split of tasks/viewpoint_select/pretrain.py:109-127."""
    no_decay = ["bias", "LayerNorm.weight"]
    named = list(model.named_parameters())
    return [
        {"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": weight_decay},
        {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
    ]
