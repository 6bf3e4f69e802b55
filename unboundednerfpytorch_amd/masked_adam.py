"""MaskedAdam with the reference's optimizer boundary (/root/reference/FourierGrid/masked_adam.py:21-75):
param groups carry 'skip_zero_grad'; `.set_pervoxel_lr(count)`; `.step()` dispatches per parameter to
adam_upd_with_perlr / masked_adam_upd / adam_upd.  Note eps is added to the UNcorrected sqrt(v) and the
bias correction is folded into a float32 step size (adam_upd_kernel.cu:72) -- this is not torch.optim.Adam.

Implementation: the single-process case of ShardedMaskedAdam (sharded_adam.py) -- same dispatch, full-shape
`exp_avg` / `exp_avg_sq` / `step` state per parameter like the reference's, and never any communication, even when
a process group happens to be initialised."""
from .sharded_adam import ShardedMaskedAdam


class MaskedAdam(ShardedMaskedAdam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, recycle_grads=False):
        """recycle_grads: see ShardedMaskedAdam (default off = the reference's contract: `.grad` survives step())"""
        super().__init__(params, lr=lr, betas=betas, eps=eps, group=None, local_only=True, recycle_grads=recycle_grads)
