"""MaskedAdam with the reference's optimizer boundary (/root/reference/FourierGrid/masked_adam.py:21-75):
param groups carry 'skip_zero_grad'; `.set_pervoxel_lr(count)`; `.step()` dispatches per parameter to
adam_upd_with_perlr / masked_adam_upd / adam_upd.  Note eps is added to the UNcorrected sqrt(v) and the
bias correction is folded into a float32 step size (adam_upd_kernel.cu:72) -- this is not torch.optim.Adam."""
import torch

from . import adam_upd_cuda


class MaskedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.99), eps=1e-8):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        self.per_lr = None
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def set_pervoxel_lr(self, count):
        assert self.param_groups[0]['params'][0].shape == count.shape
        self.per_lr = count.float() / count.max()

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            for param in group['params']:
                if param.grad is None:
                    continue
                state = self.state[param]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(param, memory_format=torch.preserve_format)
                    state['exp_avg_sq'] = torch.zeros_like(param, memory_format=torch.preserve_format)
                state['step'] += 1
                args = (state['step'], beta1, beta2, group['lr'], group['eps'])
                if self.per_lr is not None and param.shape == self.per_lr.shape:
                    adam_upd_cuda.adam_upd_with_perlr(param, param.grad, state['exp_avg'], state['exp_avg_sq'],
                                                      self.per_lr, *args)
                elif group['skip_zero_grad']:
                    adam_upd_cuda.masked_adam_upd(param, param.grad, state['exp_avg'], state['exp_avg_sq'], *args)
                else:
                    adam_upd_cuda.adam_upd(param, param.grad, state['exp_avg'], state['exp_avg_sq'], *args)
