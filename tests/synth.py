"""Deterministic, library-version-independent synthetic data for tests, golden fixtures and bench.

splitmix64 -> uniform(0,1) -> Box-Muller normal, all in numpy integer/float64 arithmetic, so the
same (seed, n) gives the same fp32 array in the build container and on the GPU box regardless of
torch/numpy RNG implementation details.  Fixtures in tests/golden store only reference OUTPUTS;
inputs are regenerated from seeds through this file.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(idx, seed):
    with np.errstate(over='ignore'):
        z = (idx.astype(np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform(seed, n, lo=0.0, hi=1.0, chunk=1 << 24):
    """n fp32 numbers in [lo,hi)."""
    out = np.empty(n, dtype=np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        bits = _splitmix64(np.arange(s, e, dtype=np.uint64), seed)
        u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        out[s:e] = (lo + (hi - lo) * u).astype(np.float32)
    return out


def normal(seed, n, mean=0.0, std=1.0, chunk=1 << 24):
    """n fp32 N(mean, std^2) numbers (Box-Muller on two decorrelated splitmix streams)."""
    out = np.empty(n, dtype=np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        idx = np.arange(s, e, dtype=np.uint64)
        b1 = _splitmix64(idx, seed)
        b2 = _splitmix64(idx, seed ^ 0x5DEECE66D)
        u1 = ((b1 >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740993.0)
        u2 = (b2 >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
        out[s:e] = (mean + std * z).astype(np.float32)
    return out


def fouriergrid_params(seed, G, F, C, width=128, depth=3, viewbase_pe=4, dens_mean=-3.0, dens_std=8.0):
    """Synthetic FourierGridModel parameters as numpy arrays (fp32), keyed like the reference
    checkpoint.  C==0 -> no rgbnet, 3-channel single-level k0."""
    P = 1 + 2 * F
    p = {'density.grid': normal(seed + 1, P * G ** 3, dens_mean, dens_std).reshape(P, 1, G, G, G)}
    if C > 0:
        p['k0.grid'] = normal(seed + 2, P * C * G ** 3).reshape(P, C, G, G, G)
        dims = [C + 3 + 6 * viewbase_pe] + [width] * (depth - 1) + [3]
        names = ['rgbnet.0'] + ['rgbnet.%d.0' % i for i in range(2, depth)] + ['rgbnet.%d' % depth]
        for li, name in enumerate(names):
            fan_in, fan_out = dims[li], dims[li + 1]
            b = 1.0 / np.sqrt(fan_in)
            p[name + '.weight'] = uniform(seed + 10 + li, fan_out * fan_in, -b, b).reshape(fan_out, fan_in)
            p[name + '.bias'] = uniform(seed + 20 + li, fan_out, -b, b)
    else:
        p['k0.grid'] = normal(seed + 2, 3 * G ** 3).reshape(1, 3, G, G, G)
    return p


def rays(seed, R, origin_scale=0.3):
    """Random camera-ish rays: origins near the scene centre, arbitrary (non-unit) directions."""
    o = normal(seed + 100, R * 3, 0.0, origin_scale).reshape(R, 3)
    d = normal(seed + 101, R * 3).reshape(R, 3)
    d = d * uniform(seed + 102, R, 0.5, 2.0).reshape(R, 1)
    v = d / np.linalg.norm(d.astype(np.float64), axis=-1, keepdims=True)
    return o.astype(np.float32), d.astype(np.float32), v.astype(np.float32)


def dvgo_params(seed, world_size, C, rgbnet_direct, viewbase_pe=4, width=128, dens_mean=8.0, dens_std=9.0):
    """Synthetic DirectVoxGO parameters keyed like the reference state dict (+ 'mask_cache.mask').
    C == 0: coarse stage (3-channel k0, no rgbnet)."""
    ws = [int(x) for x in world_size]
    n = int(np.prod(ws))
    kc = C if C > 0 else 3
    p = {'density.grid': normal(seed + 1, n, dens_mean, dens_std).reshape(1, 1, *ws),
         'k0.grid': normal(seed + 2, kc * n).reshape(1, kc, *ws),
         'mask_cache.mask': (uniform(seed + 3, n) > 0.25).reshape(*ws)}
    if C > 0:
        dim0 = 3 + 6 * viewbase_pe + (C if rgbnet_direct else C - 3)
        dims = [dim0, width, width, 3]
        for li, name in enumerate(['rgbnet.0', 'rgbnet.2.0', 'rgbnet.3']):
            b = 1.0 / np.sqrt(dims[li])
            p[name + '.weight'] = uniform(seed + 10 + li, dims[li + 1] * dims[li], -b, b).reshape(dims[li + 1], dims[li])
            p[name + '.bias'] = uniform(seed + 20 + li, dims[li + 1], -b, b)
    return p


def distortion_inputs():
    """Flattened survivor list of 9 rays (two of them empty): weights, s ascending inside each ray, sorted ray ids."""
    counts = [5, 0, 17, 1, 64, 0, 70, 3, 130]
    ray_id = np.concatenate([np.full(c, r, dtype=np.int64) for r, c in enumerate(counts)])
    n = ray_id.shape[0]
    w = uniform(301, n, 0.0, 0.08).astype(np.float32)
    s = np.concatenate([np.sort(uniform(310 + r, c, 0.0, 1.0)) for r, c in enumerate(counts) if c > 0]).astype(np.float32)
    return w, s, ray_id, 256


# the training-step golden (tests/golden/train_step.npz): model and batch
TRAIN_CASE = dict(seed=21, G=12, F=3, C=12, pe=4, norm="inf", thres=1e-4, dm=4.0, ds=10.0, R=96, stepsize=0.5)


# the Fourier-loss golden (tests/golden/fourier_loss.npz): bicycle_single.py:46-57's loss weights (those run_train.py evaluates without
# the third-party distortion package) on TRAIN_CASE, and the near-clip threshold used there
FREQ_WEIGHTS = dict(weight_main=1.0, weight_freq=5.0, weight_entropy_last=0.001, weight_nearclip=1.0)
FREQ_NEAR = 0.35


# dcvgo.DirectContractedVoxGO goldens: name, seed, G (num_voxels = G^3), Gb (num_voxels_base), C (0 = coarse, 3-channel
# k0 without rgbnet), contracted_norm, rays, density mean / std
DCVGO_CASES = [
    ("dcvgo_fine_inf", 51, 14, 16, 12, "inf", 90, 1.0, 5.0),
    ("dcvgo_coarse_l2", 52, 12, 12, 0, "l2", 80, 0.0, 4.0),
]
DCVGO_BOX = ([-0.9, -1.1, -1.0], [1.1, 0.9, 1.0])     # the fg/bg separating cube (must be a cube), off the origin


def dvgo_views():
    """Three small pinhole views around the dvgo test box: H, W, K [3,3], poses [3][3,4] (camera-to-world)."""
    H, W = 12, 16
    K = np.array([[14.0, 0, W / 2], [0, 14.0, H / 2], [0, 0, 1]], dtype=np.float32)
    poses = []
    for a, h in ((0.3, 0.2), (2.2, -0.3), (4.0, 0.5)):
        eye = np.array([2.6 * np.cos(a), 2.6 * np.sin(a), h], dtype=np.float64)
        fwd = -eye / np.linalg.norm(eye)
        right = np.cross(fwd, [0.0, 0.0, 1.0]); right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        poses.append(np.stack([right, up, -fwd, eye], axis=1).astype(np.float32))    # OpenGL: x right, y up, looks along -z
    return H, W, K, poses


# model-level training utilities golden (tests/golden/fg_model_utils.npz): a sparse scene so that the occupancy
# cache really changes; scale_volume_grid goes from G^3 to G2^3 voxels
MODEL_UTILS_CASE = dict(seed=61, G=8, G2=10, F=2, C=4, pe=2, norm="inf", thres=1e-4, dm=-1.8, ds=3.0)



# ---------------------------------------------------------------------------------------------------------------------
# Bounds of the two native-step tests (tests/test_gpu_train_scale.py, tests/test_gpu_voxgo_train.py), set from the committed
# distribution profiles/r06/native_step_spread.json (tools/native_step_spread.py: 100 repetitions x 4 configurations of native vs
# op-by-op AND op-by-op vs op-by-op -- the two distributions coincide: what differs between two runs is the order of the grid
# scatters' fp32 atomics, in either path).  Every bound is <= 4 x the largest value observed in those 400 repetitions:
#   grid gradients         max |dA - dB| / max |dB|      observed 2.14e-6 (op vs op 1.80e-6)     -> 8e-6      (round 4: 2e-6, which the
#                                                                                                   noise exceeds in ~2 of 100 runs)
#   loss trajectory        max relative difference        observed 1.19e-7 (op vs op 1.04e-7)     -> 5e-7
#   parameters after the short trajectories: largest difference 0.206 of a learning-rate step (op vs op 0.206) -> 0.85; entries further
#   apart than 2 % of a step: at most 13 of 4 992 (2.6e-3 of a tensor) and 58 of 108 M, in 5 of 100 runs (op vs op: 8 of 100) --
#   Adam's first steps are sign-like, an entry whose gradient is within rounding of zero moves by +-lr in either run
#                                                                                                 -> max(4, min(1.05e-2 numel, 232))
# Forward arrays, loss, mse and the fixed-order (rgbnet) gradients were bit-identical in all 400: the tests keep torch.equal there.
# ---------------------------------------------------------------------------------------------------------------------
NATIVE_GRID_GRAD_BOUND = 8e-6
NATIVE_LOSS_RTOL = 5e-7
NATIVE_PARAM_MAX_LR_STEPS = 0.85


def native_param_outlier_limit(numel):
    return max(4, min(int(1.05e-2 * numel), 232))


def assert_same_trajectory(params_a, params_b, lr_of=lambda name: 0.1 if "grid" in name else 1e-3):
    """the parameter dicts of two short training runs of the same step (tensors), under the bounds above"""
    for k in params_a:
        diff = (params_a[k] - params_b[k]).abs()
        lr = lr_of(k)
        worst, n_out = float(diff.max()) / lr, int((diff > 0.02 * lr).sum())
        assert worst <= NATIVE_PARAM_MAX_LR_STEPS, (k, worst)
        assert n_out <= native_param_outlier_limit(diff.numel()), (k, n_out, diff.numel(), worst)
