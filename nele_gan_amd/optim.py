"""torch.optim.Adam semantics (train_nele.py:89-91) as ONE fused HIP pass over a model's flat
parameter / gradient buffers (csrc/disc.hip: adam_kernel)."""
import torch

from . import ops


class Adam:
    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if not hasattr(module, 'flat_parameters'):
            raise TypeError("nele_gan_amd.optim.Adam takes a nele_gan_amd.model module (flat parameter buffer)")
        self.module = module
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.step_count = 0
        self.m = None
        self.v = None
        self.guard = None        # device int[2]: {last step whose gradient was not finite, number of skipped steps}

    def _state(self):
        fp = self.module.flat_parameters()
        if self.m is None or self.m.device != fp.flat.device or self.m.numel() != fp.flat.numel():
            self.m = torch.zeros_like(fp.flat)
            self.v = torch.zeros_like(fp.flat)
            self.guard = torch.zeros(2, dtype=torch.int32, device=fp.flat.device)
        return fp

    def zero_grad(self):
        self._state().grad.zero_()

    def step(self):
        """One Adam update.  A step whose gradient is not finite is skipped on the device (parameters and moments untouched) and does not
        count: the bias correction of later steps uses the number of updates that happened (step_count - skipped, read on the device), as
        torch.optim.Adam would after a loop that simply did not call step() for that batch."""
        fp = self._state()
        self.step_count += 1
        ops.adam_step(fp.flat, fp.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.step_count, self.guard)

    def skipped_steps(self):
        """Optimiser steps masked on the device because the gradient was not finite (synchronises the host)."""
        return 0 if self.guard is None else int(self.guard[1].item())

    def state_dict(self):
        self._state()
        return {'step': self.step_count, 'exp_avg': self.m, 'exp_avg_sq': self.v, 'lr': self.lr, 'betas': self.betas, 'eps': self.eps,
                'skipped': self.skipped_steps()}

    def load_state_dict(self, sd):
        self._state()
        self.step_count = int(sd['step'])
        self.m.copy_(sd['exp_avg'])
        self.v.copy_(sd['exp_avg_sq'])
        # guard[0] holds the step NUMBER of the last non-finite gradient: after rewinding step_count a stale value would mask a finite
        # step of the same number (and let a bad one through).  The skipped-step count survives in the state dict.
        self.guard.zero_()
        if 'skipped' in sd:
            self.guard[1] = int(sd['skipped'])
