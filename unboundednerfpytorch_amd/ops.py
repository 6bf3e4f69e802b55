"""Autograd boundary of the hot path, same class names / argument order / return arity as the
reference (/root/reference/FourierGrid/dvgo.py:430-488): Raw2Alpha, Raw2Alpha_nonuni, Alphas2Weights.
All three are once_differentiable and save tensors only when the input requires grad.
DistortionLoss mirrors FourierGrid_model.py:684-708 (the mip-NeRF-360 distortion regulariser over the flattened
survivor list) INCLUDING that dead-code class's quirk (backward not divided by n_rays).  The training loop itself
(run_train.py:270-275) calls the third-party torch_efficient_distloss.flatten_eff_distloss, whose forward is the
same value and whose backward IS the derivative of that forward (divided by n_rays): FlattenEffDistLoss /
flatten_eff_distloss below, the default of train_step.training_loss."""
import torch

from . import _lib, render_utils_cuda, ub360_utils_cuda

_L = _lib.load()


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


class DistortionLoss(torch.autograd.Function):
    """loss = (1/n_rays) * sum over samples of [ 2 w_i (s_i W_<i - WS_<i)  +  w_i^2 / (3 n_max) ], with W_<i / WS_<i
    the exclusive per-ray running sums of w and w*s (HIP segment_cumsum).  For s ascending inside a ray this is
    sum_{i != j} w_i w_j |s_i - s_j| plus the intra-interval term of width 1/n_max.  Gradient w.r.t. w only.
    Reference quirk kept: the forward divides by n_rays, the backward does not (FourierGrid_model.py:699-708)."""

    @staticmethod
    def forward(ctx, w, s, n_max, ray_id):
        n_rays = ray_id.max() + 1
        width = 1 / n_max
        cumsum = DistortionLoss.segment_cumsum or ub360_utils_cuda.segment_cumsum
        w_pre, w_tot, ws_pre, ws_tot = cumsum(w, s, ray_id, int(n_rays))
        pair = 2 * w * (s * w_pre - ws_pre)
        self_term = (1 / 3) * width * w.pow(2)      # same operation order as the reference's loss_uni
        ctx.save_for_backward(w, s, w_pre, w_tot, ws_pre, ws_tot, ray_id)
        ctx.width = width
        ctx.n_rays = n_rays
        return (pair.sum() + self_term.sum()) / n_rays

    normalize_backward = False   # the reference's dead-code class forgets the 1/n_rays of its own forward
    segment_cumsum = None        # test hook: another implementation of the op (the oracle's); None = the HIP kernel

    @classmethod
    def _grad(cls, ctx, grad_back):
        w, s, w_pre, w_tot, ws_pre, ws_tot, ray_id = ctx.saved_tensors
        # sums over the samples AFTER i in the same ray: total - (prefix + own)
        w_after = w_tot[ray_id] - (w_pre + w)
        ws_after = ws_tot[ray_id] - (ws_pre + w * s)
        d_pair = 2 * (s * (w_pre - w_after) + (ws_after - ws_pre))
        d_self = (1 / 3) * ctx.width * 2 * w
        grad = grad_back * (d_pair + d_self)
        return grad / ctx.n_rays if cls.normalize_backward else grad

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_back):
        return DistortionLoss._grad(ctx, grad_back), None, None, None


class FlattenEffDistLoss(DistortionLoss):
    """flatten_eff_distloss(w, m, interval, ray_id) of torch_efficient_distloss, the function run_train.py:274 calls
    (third-party, not vendored by the reference; restated from its published algorithm): same forward as
    DistortionLoss with `interval` given directly (the loop passes 1/n_max), and a backward that is the exact
    derivative of the forward, i.e. divided by n_rays = ray_id.max()+1 (checked by gradcheck in fp64 on the oracle
    op, tests/test_host_logic.py)."""
    normalize_backward = True

    @staticmethod
    def forward(ctx, w, m, interval, ray_id):
        return DistortionLoss.forward(ctx, w, m, 1 / interval, ray_id)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_back):
        return FlattenEffDistLoss._grad(ctx, grad_back), None, None, None


distortion_loss = DistortionLoss.apply          # (w, s, n_max, ray_id): the reference's in-repo class, quirk kept
flatten_eff_distloss = FlattenEffDistLoss.apply  # (w, m, interval, ray_id): what the training loop uses


class SplitKLinear(torch.autograd.Function):
    """y = x W^T + b for a tall x [M, K] (M = the step's survivors, ~1e5) and a small W [N, K].  Same forward as
    nn.Linear (FourierGrid_model.py:636 calls the layers of self.rgbnet); the weight gradient go^T x has a 128 x 128 (or
    3 x 128) result and an M-long reduction, which rocBLAS runs on 16 workgroups (185-225 us at M = 84k): here the rows
    are cut into SPLIT slabs reduced by one batched GEMM + a sum (55 us).  fp32 throughout; the summation order of the
    weight gradient differs from the library's single GEMM (rel. 4e-6, tools/microbench/rgbnet_gemms.py)."""
    SPLIT = 32

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go):
        x, weight = ctx.saved_tensors
        go = go.contiguous()
        gx = go @ weight if ctx.needs_input_grad[0] else None
        M, S = go.shape[0], SplitKLinear.SPLIT
        m = M // S
        main = m * S
        if m >= 64:
            gw = torch.bmm(go[:main].view(S, m, -1).transpose(1, 2), x[:main].view(S, m, -1)).sum(0)
            if main < M:
                gw = gw.addmm_(go[main:].t(), x[main:])
        else:
            gw = go.t() @ x
        return gx, gw, go.sum(0)


def rgbnet_linears(net):
    """the three nn.Linear layers of an rgbnet of the reference's default depth (Linear(k,W)-ReLU-[Linear(W,W)-ReLU]-Linear(W,3),
    W <= 128, k <= 128: FourierGrid_model.py:233-241) or None for any other network (depth, wider, extra layers)"""
    lin, other = [], []
    for m in net.modules():
        if isinstance(m, torch.nn.Linear):
            lin.append(m)
        elif not isinstance(m, (torch.nn.Sequential, torch.nn.ReLU)):
            other.append(m)
    if other or len(lin) != 3 or any(l.bias is None for l in lin):
        return None
    W = lin[0].out_features
    if not (1 <= W <= 128) or lin[1].in_features != W or lin[1].out_features != W or lin[2].in_features != W \
            or lin[2].out_features != 3 or lin[0].in_features > 128:
        return None
    return lin


class ViewRows(tuple):
    """(viewdirs [N,3], viewfreq [pe], ray_id [M]): the view embedding of the M samples, not yet formed.  FusedRgbnet takes it in
    place of the embedding rows and builds the rgbnet's input in one launch (rgbnet_features)."""

    def __new__(cls, viewdirs, viewfreq, ray_id):
        return super().__new__(cls, (viewdirs, viewfreq, ray_id))


def rgbnet_features(k0, viewdirs, viewfreq, ray_id):
    """cat([k0, emb], -1) with emb = cat([viewdirs, e.sin(), e.cos()], -1)[ray_id], e = (viewdirs[..., None] * viewfreq).flatten(-2)
    (FourierGrid_model.py:631-635, dvgo.py:352-357) -- on the GPU one kernel (include/ugrid_hip.h: ugrid_rgbnet_features) in place
    of that chain's six elementwise launches and two concatenations.  k0 [M,C] or None (the embedding rows alone); viewdirs
    [..., 3] (flattened to rays); ray_id [M] or None.  No gradient flows (k0's derivative is FusedRgbnet's business)."""
    viewdirs = viewdirs.reshape(-1, 3)
    if not viewdirs.is_cuda:
        e = (viewdirs.unsqueeze(-1) * viewfreq).flatten(-2)
        emb = torch.cat([viewdirs, e.sin(), e.cos()], -1)
        emb = emb if ray_id is None else emb[ray_id]
        return emb if k0 is None else torch.cat([k0, emb], -1)
    viewdirs, viewfreq = viewdirs.contiguous(), viewfreq.contiguous()
    named = [("viewdirs", viewdirs), ("viewfreq", viewfreq)]
    if k0 is not None:
        k0 = k0.detach().contiguous()
        named.append(("k0", k0))
    _lib.require_cuda(*named, *([("ray_id", ray_id)] if ray_id is not None else []))
    _lib.require_f32(*named)
    if ray_id is not None:
        if ray_id.dtype != torch.int64:
            raise TypeError("rgbnet_features: ray_id must be int64, got %s" % ray_id.dtype)
        ray_id = ray_id.contiguous()
    M = viewdirs.shape[0] if ray_id is None else ray_id.shape[0]
    C = 0 if k0 is None else k0.shape[1]
    if k0 is not None and k0.shape[0] != M:
        raise ValueError("rgbnet_features: k0 has %d rows, the samples are %d" % (k0.shape[0], M))
    pe = viewfreq.numel()
    out = torch.empty(M, C + 3 + 6 * pe, device=viewdirs.device)
    N = viewdirs.shape[0]
    # many samples per ray: the embedding is formed once per ray and gathered (the library decides; same values either way)
    ray_rows = torch.empty(N, 3 + 6 * pe, device=viewdirs.device) if (ray_id is not None and M >= 2 * N) else None
    with _lib.guard(out.device):
        _lib.check(_L.ugrid_rgbnet_features(_lib.ptr(k0) if C else None, C, _lib.ptr(viewdirs), N, _lib.ptr(viewfreq) if pe else None, pe,
                                            _lib.ptr(ray_id) if ray_id is not None else None, M, _lib.ptr(ray_rows), _lib.ptr(out),
                                            _lib.stream_of(out)), "rgbnet_features")
    return out


class FusedRgbnet(torch.autograd.Function):
    """logits = rgbnet(cat([k0, emb])) for the default 3 x 128 rgbnet, forward and backward on the hand-written fp32-MFMA
    kernels of csrc/ugrid_train_mlp.hip (include/ugrid_hip.h: ugrid_rgbnet_train_forward / _backward) instead of 13 library
    GEMMs + elementwise kernels.  k0 [M,C] (gradient returned), emb [M,E] (view embedding rows, no gradient) or a ViewRows
    (the rows are then formed together with the concatenation, one launch), then the three
    weights and biases in nn.Linear layout.  fp32; deterministic (the weight gradients are fixed-order sums of slab partials)."""

    @staticmethod
    def forward(ctx, k0, emb, w0, b0, w1, b1, w2, b2):
        ws = [t.contiguous() for t in (w0, b0, w1, b1, w2, b2)]
        if isinstance(emb, ViewRows):
            _lib.require_cuda(("k0", k0), *[("rgbnet", t) for t in ws])
            _lib.require_f32(("k0", k0), *[("rgbnet", t) for t in ws])
            feat = rgbnet_features(k0, *emb)
        else:
            feat = torch.cat([k0, emb], -1).contiguous()
            _lib.require_cuda(("k0", k0), ("emb", emb), *[("rgbnet", t) for t in ws])
            _lib.require_f32(("k0", k0), ("emb", emb), *[("rgbnet", t) for t in ws])
        M, K = feat.shape
        dev = feat.device
        W = ws[0].shape[0]
        h1 = torch.empty(M, W, device=dev)
        h2 = torch.empty(M, W, device=dev)
        logits = torch.empty(M, 3, device=dev)
        with _lib.guard(dev):
            _lib.check(_L.ugrid_rgbnet_train_forward(_lib.ptr(feat), M, K, *[_lib.ptr(t) for t in ws], W, _lib.ptr(h1), _lib.ptr(h2),
                                                     _lib.ptr(logits), _lib.stream_of(feat)), "rgbnet_train_forward")
        ctx.save_for_backward(feat, h1, h2, ws[0], ws[2], ws[4])
        ctx.C = k0.shape[1]
        ctx.need_k0 = k0.requires_grad
        return logits

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_logits):
        feat, h1, h2, w0, w1, w2 = ctx.saved_tensors
        M, K = feat.shape
        dev = feat.device
        g_logits = g_logits.to(torch.float32).contiguous()
        n_fg = ctx.C if ctx.need_k0 else 0
        g_k0 = torch.empty(M, n_fg, device=dev) if n_fg else None
        W = w0.shape[0]
        g = [torch.empty_like(w0), torch.empty(W, device=dev), torch.empty_like(w1), torch.empty(W, device=dev),
             torch.empty_like(w2), torch.empty(3, device=dev)]
        scratch = torch.empty(int(_L.ugrid_rgbnet_train_scratch_floats(M)), device=dev)
        with _lib.guard(dev):
            _lib.check(_L.ugrid_rgbnet_train_backward(_lib.ptr(g_logits), _lib.ptr(feat), _lib.ptr(h1), _lib.ptr(h2), M, K, n_fg,
                                                      _lib.ptr(w0), _lib.ptr(w1), _lib.ptr(w2), W, _lib.ptr(g_k0) if n_fg else None,
                                                      *[_lib.ptr(t) for t in g], _lib.ptr(scratch), _lib.stream_of(feat)),
                       "rgbnet_train_backward")
        return (g_k0, None, *g)


def sequential_splitk(net, x):
    """Applies an nn.Sequential of Linear / ReLU / nested Sequential (the rgbnet) with SplitKLinear for the Linear layers."""
    for layer in net:
        if isinstance(layer, torch.nn.Linear):
            x = SplitKLinear.apply(x, layer.weight, layer.bias)
        elif isinstance(layer, torch.nn.ReLU):
            x = torch.relu_(x) if layer.inplace else torch.relu(x)
        elif isinstance(layer, torch.nn.Sequential):
            x = sequential_splitk(layer, x)
        else:
            x = layer(x)
    return x


class RenderLoss(torch.autograd.Function):
    """Training tail in one op (include/ugrid_hip.h: ugrid_render_loss): rgb = sigmoid(logits), rgb_marched = per-ray
    sum of weights * rgb + alphainv_last * bg, and the loss of run_train.py:254-279 (main MSE, the image-space Fourier MSE of
    FourierGrid_model.py:112-129, entropy_last, nearclip, flatten_eff_distloss with interval = 1 / n_max, rgbper) -- what fourier_model.FourierGridModel.forward's tail and
    train_step.training_loss compute with ~45 torch launches (and ~90 in the backward).  ray_id must be ascending.

    forward(logits [M,3], weights [M], alphainv_last [R], raw_density [M], ray_id [M], t [M], s [M] or None (then
            s = 1 - 1/(1+t)), target [R,3], bg [R,3] or None, coef) -> loss (scalar), mse (scalar, no gradient), rgb_marched [R,3] (no gradient)
    coef = (weight_main, weight_entropy_last, weight_distortion, weight_rgbper, weight_nearclip * world_size, near_thres,
            interval, n_rays, weight_freq); gradients flow to logits, weights, alphainv_last and raw_density."""

    @staticmethod
    def forward(ctx, logits, weights, alphainv_last, raw_density, ray_id, t, s, target, bg, coef):
        import ctypes
        named = [("logits", logits), ("weights", weights), ("alphainv_last", alphainv_last), ("t", t), ("target", target)]
        if s is not None:
            named.append(("s", s))
        _lib.require_cuda(*named, ("ray_id", ray_id))
        _lib.require_f32(*named)
        M, R = weights.shape[0], alphainv_last.shape[0]
        if logits.shape != (M, 3) or target.shape != (R, 3) or ray_id.dtype != torch.int64 or (bg is not None and bg.shape != (R, 3)):
            raise RuntimeError("RenderLoss: logits [M,3], weights [M], ray_id int64 [M], alphainv_last [R], target / bg [R,3]")
        dev = logits.device
        logits, weights, alphainv_last, t, target = (x.contiguous() for x in (logits, weights, alphainv_last, t, target))
        s = s.contiguous() if s is not None else None
        bg = bg.contiguous() if bg is not None else None
        ray_id = ray_id.contiguous()
        c9 = [float(x) for x in coef]
        if len(c9) not in (8, 9):
            raise RuntimeError("RenderLoss: coef holds 8 or 9 numbers (ops.loss_coefficients)")
        h = (ctypes.c_float * 9)(*(c9 + [0.0] * (9 - len(c9))))
        seg = torch.empty(2 * R, dtype=torch.int64, device=dev)
        rgb_marched = torch.empty(R, 3, device=dev)
        ray_tot = torch.empty(R, 2, device=dev)
        partial = torch.empty(R, 5, device=dev)
        out2 = torch.empty(2, device=dev)
        with _lib.guard(dev):
            _lib.check(_L.ugrid_render_loss(_lib.ptr(logits), _lib.ptr(weights), _lib.ptr(s) if s is not None else None, _lib.ptr(t),
                                            _lib.ptr(alphainv_last),
                                            _lib.ptr(bg) if bg is not None else None, _lib.ptr(target), _lib.ptr(ray_id), M, R,
                                            ctypes.cast(h, ctypes.c_void_p), _lib.ptr(seg), _lib.ptr(rgb_marched), _lib.ptr(ray_tot),
                                            _lib.ptr(partial), _lib.ptr(out2), _lib.stream_of(logits)), "render_loss")
        ctx.save_for_backward(logits, weights, alphainv_last, ray_id, t, target, seg, rgb_marched, ray_tot)
        ctx.bg, ctx.h, ctx.s = bg, h, s
        loss, mse = out2[0], out2[1]
        ctx.mark_non_differentiable(mse, rgb_marched)
        return loss, mse, rgb_marched

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss, g_mse, g_rm):
        import ctypes
        logits, weights, alphainv_last, ray_id, t, target, seg, rgb_marched, ray_tot = ctx.saved_tensors
        s = ctx.s
        M, R = weights.shape[0], alphainv_last.shape[0]
        g_loss = g_loss.to(torch.float32).reshape(1).contiguous()
        g_logits, g_w, g_ainv, g_dens = torch.empty_like(logits), torch.empty_like(weights), torch.empty_like(alphainv_last), \
            torch.empty_like(weights)
        with _lib.guard(logits.device):
            _lib.check(_L.ugrid_render_loss_backward(
                _lib.ptr(logits), _lib.ptr(weights), _lib.ptr(s) if s is not None else None, _lib.ptr(t), _lib.ptr(alphainv_last),
                _lib.ptr(ctx.bg) if ctx.bg is not None else None, _lib.ptr(target), _lib.ptr(ray_id), M, R,
                ctypes.cast(ctx.h, ctypes.c_void_p), _lib.ptr(seg), _lib.ptr(rgb_marched), _lib.ptr(ray_tot), _lib.ptr(g_loss),
                _lib.ptr(g_logits), _lib.ptr(g_w), _lib.ptr(g_ainv), _lib.ptr(g_dens), _lib.stream_of(logits)), "render_loss_backward")
        return g_logits, g_w, g_ainv, g_dens, None, None, None, None, None, None


def loss_coefficients(cfg_train, n_rays, n_max, near_thres=None, world_size=1):
    """the coef tuple of RenderLoss from a cfg_train (dict or attribute object, run_train.py:254-279), or None when the
    configuration cannot be evaluated by the fused op (nearclip without its threshold).  weight_freq -- the image-space Fourier
    loss of bicycle_single.py:57 / stump_single.py:55 (5.0), barn / caterpillar (0.3), waymo_no_block (1.0) -- is the 9th entry."""
    get = (lambda k, d=0.0: cfg_train.get(k, d)) if isinstance(cfg_train, dict) else (lambda k, d=0.0: getattr(cfg_train, k, d))
    w_near = get('weight_nearclip', 0.0)
    if w_near > 0 and near_thres is None:
        return None
    return (get('weight_main', 1.0), max(get('weight_entropy_last', 0.0), 0.0), max(get('weight_distortion', 0.0), 0.0),
            max(get('weight_rgbper', 0.0), 0.0), (w_near * world_size) if w_near > 0 else 0.0,
            near_thres if near_thres is not None else 0.0, 1.0 / n_max, float(n_rays), float(get('weight_freq', 0.0) or 0.0))
