"""RAdam with the reference's exact update rule (utils/optim/radam.py:6-98 of the reference): rectified Adam with an
SGD-style fallback while N_sma < 5 (the first 5 steps at beta2 = 0.999).  Device tensors are updated by the fused HIP
kernel kd_radam_step (one launch per tensor, no fp32 round-trip copies, no host sync); CPU tensors (the CIFAR plumbing
config) take an equivalent torch path.  The reference's 10-slot (step -> N_sma, step_size) cache is a pure function of
`step`, recomputed here (same values)."""
import math

import torch
from torch.optim.optimizer import Optimizer

from ... import ops


def _rect(step, beta1, beta2, degenerated_to_sgd=True):
    beta2_t = beta2 ** step
    n_max = 2 / (1 - beta2) - 1
    n_sma = n_max - 2 * step * beta2_t / (1 - beta2_t)
    if n_sma >= 5:
        step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2)) / \
            (1 - beta1 ** step)
    elif degenerated_to_sgd:
        step_size = 1.0 / (1 - beta1 ** step)
    else:
        step_size = -1
    return n_sma, step_size


class RAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, degenerated_to_sgd=True):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        self.degenerated_to_sgd = degenerated_to_sgd
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        batch = []   # device tensors: one multi-tensor launch for all of them (kd_radam_step_multi)
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError('RAdam does not support sparse gradients')
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
                state['step'] += 1
                step = state['step']
                if p.is_cuda and p.dtype == torch.float32 and p.is_contiguous():
                    if not self.degenerated_to_sgd and _rect(step, beta1, beta2, False)[1] < 0:
                        # no parameter update this step, moments still advance (reference behaviour)
                        g = p.grad.float()
                        state['exp_avg_sq'].mul_(beta2).addcmul_(g, g, value=1 - beta2)
                        state['exp_avg'].mul_(beta1).add_(g, alpha=1 - beta1)
                        continue
                    g = p.grad if (p.grad.dtype == torch.float32 and p.grad.is_contiguous()) else p.grad.float().contiguous()
                    batch.append((p, g, state['exp_avg'], state['exp_avg_sq'], step, group['lr'], beta1, beta2,
                                  group['eps'], group['weight_decay']))
                    continue
                # torch path (CPU tensors)
                grad = p.grad.float()
                p32 = p.float()
                exp_avg, exp_avg_sq = state['exp_avg'], state['exp_avg_sq']
                exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
                exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1)
                n_sma, step_size = _rect(step, beta1, beta2, self.degenerated_to_sgd)
                if n_sma >= 5:
                    if group['weight_decay'] != 0:
                        p32.add_(p32, alpha=-group['weight_decay'] * group['lr'])
                    p32.addcdiv_(exp_avg, exp_avg_sq.sqrt().add_(group['eps']), value=-step_size * group['lr'])
                    p.copy_(p32)
                elif step_size > 0:
                    if group['weight_decay'] != 0:
                        p32.add_(p32, alpha=-group['weight_decay'] * group['lr'])
                    p32.add_(exp_avg, alpha=-step_size * group['lr'])
                    p.copy_(p32)
        if len(batch) == 1:
            ops.radam_step(*batch[0])
        elif batch:
            ops.radam_step_multi(batch)
        return loss
