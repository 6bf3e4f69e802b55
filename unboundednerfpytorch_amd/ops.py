"""Autograd boundary of the hot path, same class names / argument order / return arity as the
reference (/root/reference/FourierGrid/dvgo.py:430-488): Raw2Alpha, Raw2Alpha_nonuni, Alphas2Weights.
All three are once_differentiable and save tensors only when the input requires grad."""
import torch

from . import render_utils_cuda


class Raw2Alpha(torch.autograd.Function):
    """alpha = 1 - (1 + exp(density + shift)) ** (-interval)"""

    @staticmethod
    def forward(ctx, density, shift, interval):
        exp, alpha = render_utils_cuda.raw2alpha(density, shift, interval)
        if density.requires_grad:
            ctx.save_for_backward(exp)
            ctx.interval = interval
        return alpha

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_back):
        (exp,) = ctx.saved_tensors
        return render_utils_cuda.raw2alpha_backward(exp, grad_back.contiguous(), ctx.interval), None, None


class Raw2Alpha_nonuni(torch.autograd.Function):
    """Raw2Alpha with a per-point interval tensor."""

    @staticmethod
    def forward(ctx, density, shift, interval):
        exp, alpha = render_utils_cuda.raw2alpha_nonuni(density, shift, interval)
        if density.requires_grad:
            ctx.save_for_backward(exp)
            ctx.interval = interval
        return alpha

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_back):
        (exp,) = ctx.saved_tensors
        return render_utils_cuda.raw2alpha_nonuni_backward(exp, grad_back.contiguous(), ctx.interval), None, None


class Alphas2Weights(torch.autograd.Function):
    """(alpha[n], ray_id[n] sorted, N rays) -> (weights[n], alphainv_last[N])"""

    @staticmethod
    def forward(ctx, alpha, ray_id, N):
        weights, T, alphainv_last, i_start, i_end = render_utils_cuda.alpha2weight(alpha, ray_id, N)
        if alpha.requires_grad:
            ctx.save_for_backward(alpha, weights, T, alphainv_last, i_start, i_end)
            ctx.n_rays = N
        return weights, alphainv_last

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_weights, grad_last):
        alpha, weights, T, alphainv_last, i_start, i_end = ctx.saved_tensors
        grad = render_utils_cuda.alpha2weight_backward(
            alpha, weights, T, alphainv_last, i_start, i_end, ctx.n_rays,
            grad_weights.contiguous(), grad_last.contiguous())
        return grad, None, None
