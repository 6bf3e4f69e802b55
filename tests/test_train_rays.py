"""Ray-batch side of the training loop (unboundednerfpytorch_amd/train_rays.py) against golden vectors produced by the
reference's own functions (tests/golden/gen_train_rays_golden.py): the flattened / image-shaped ray tables, the FourierGrid
variant with image indices and pose refinement, the numpy-stream batch sampler, the 'random' mode's torch draws, and
gather_training_rays / sample_batch end to end.  CPU: bitwise.  GPU (one-kernel ray generation, resident table): 1e-6."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
GOLD = os.path.join(ROOT, "tests", "golden", "train_rays.npz")
FLAGS = dict(ndc=False, inverse_y=True, flip_x=False, flip_y=True)


def scene(device="cpu"):
    import gen_train_rays_golden as gen
    imgs, poses, HW, Ks = gen.scene()
    return [im.to(device) for im in imgs], poses, HW, Ks


def check(a, b, exact):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    if exact:
        np.testing.assert_array_equal(a, b)
    else:
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-6)


def run_tables(device, exact):
    from unboundednerfpytorch_amd import train_rays as tr
    g = np.load(GOLD)
    imgs, poses, HW, Ks = scene(device)
    r = tr.get_training_rays_flatten(rgb_tr_ori=imgs, train_poses=poses, HW=HW, Ks=Ks, **FLAGS)
    for k, v in zip(("rgb", "o", "d", "v"), r[:4]):
        check(v, g["flat_" + k], exact)
        assert v.device.type == torch.device(device).type
    assert list(r[4]) == list(g["flat_imsz"])
    r = tr.FourierGrid_get_training_rays(rgb_tr_ori=imgs, train_poses=poses.clone(), HW=HW, Ks=Ks, **FLAGS)
    for k, v in zip(("rgb", "o", "d", "v", "idx"), r[:5]):
        check(v, g["fg_" + k], exact)
    assert list(r[5]) == list(g["fg_imsz"])
    p2 = poses.clone()
    r = tr.FourierGrid_get_training_rays(rgb_tr_ori=imgs, train_poses=p2, HW=HW, Ks=Ks, pos_emb=torch.tensor([0.1, -0.2, 0.05]), **FLAGS)
    check(r[1], g["fgpos_o"], exact)
    check(p2, g["fgpos_poses_after"], True)           # refined in place, like the reference
    same = torch.stack([imgs[0], imgs[2]])
    r = tr.get_training_rays(rgb_tr=same, train_poses=poses[[0, 2]], HW=HW[[0, 2]], Ks=np.stack([Ks[0], Ks[0]]),
                             ndc=False, inverse_y=False, flip_x=True, flip_y=False)
    for k, v in zip(("rgb", "o", "d", "v"), r[:4]):
        check(v, g["img_" + k], exact)
    assert r[4] == [1, 1]
    with pytest.raises(NotImplementedError):
        tr.get_training_rays_flatten(rgb_tr_ori=imgs, train_poses=poses, HW=HW, Ks=Ks, ndc=True, inverse_y=False, flip_x=False, flip_y=False)


def test_ray_tables_equal_the_reference_functions_on_cpu():
    run_tables("cpu", exact=True)


@pytest.mark.gpu
def test_ray_tables_on_the_gpu_resident_table():
    run_tables("cuda:0", exact=False)


def test_batch_sampler_and_random_mode_follow_the_reference_streams():
    from unboundednerfpytorch_amd import train_rays as tr
    g = np.load(GOLD)
    np.random.seed(11)
    gen = tr.batch_indices_generator(50, 16)
    got = np.stack([next(gen).numpy() for _ in range(7)])
    np.testing.assert_array_equal(got, g["sampler_batches"])
    assert len(set(got[:3].flatten().tolist())) == 48          # an epoch: no repeats, the incomplete tail is dropped
    rgb_img, o_img = torch.from_numpy(g["img_rgb"]), torch.from_numpy(g["img_o"])
    flat = torch.from_numpy(g["flat_rgb"])
    cfg = {"ray_sampler": "random", "N_rand": 10}
    torch.manual_seed(5)
    t, o, d, v, idx = tr.sample_batch(cfg, rgb_img, o_img, o_img, o_img, None, None)
    np.testing.assert_array_equal(t.numpy(), g["rand3_target"])
    np.testing.assert_array_equal(o.numpy(), g["rand3_o"])
    assert idx is None
    t, *_ = tr.sample_batch(cfg, flat, flat, flat, flat, None, None)
    np.testing.assert_array_equal(t.numpy(), g["rand2_target"])
    np.testing.assert_array_equal(torch.rand(3).numpy(), g["rand_next"])       # same number of draws as run_train.py


def test_gather_training_rays_and_sample_batch_end_to_end():
    from unboundednerfpytorch_amd import train_rays as tr
    g = np.load(GOLD)
    imgs, poses, HW, Ks = scene()
    cfg = types.SimpleNamespace(model="FourierGrid", data=types.SimpleNamespace(
        dataset_type="tankstemple", load2gpu_on_the_fly=True, ndc=False, inverse_y=True, flip_x=False, flip_y=True))
    cfg_train = types.SimpleNamespace(ray_sampler="flatten", N_rand=32)
    np.random.seed(2)
    out = tr.gather_training_rays(types.SimpleNamespace(pos_emb=None), {"irregular_shape": True}, imgs, cfg, [0, 1, 2], cfg_train,
                                  poses, HW, Ks, {}, device="cpu")
    rgb_tr, o_tr, d_tr, v_tr, idx_tr, imsz, sampler = out
    check(rgb_tr, g["fg_rgb"], True), check(o_tr, g["fg_o"], True), check(idx_tr, g["fg_idx"], True)
    np.random.seed(2)
    ref_gen = tr.batch_indices_generator(len(rgb_tr), 32)
    ref_batches = [next(ref_gen) for _ in range(6)]           # (the permutation is drawn at the first next(), not at creation)
    np.random.seed(2)
    for sel in ref_batches:
        t, o, d, v, idx = tr.sample_batch(cfg_train, rgb_tr, o_tr, d_tr, v_tr, idx_tr, sampler, device="cpu", load2gpu_on_the_fly=True)
        assert torch.equal(t, rgb_tr[sel]) and torch.equal(o, o_tr[sel]) and torch.equal(v, v_tr[sel]) and torch.equal(idx, idx_tr[sel])
    # non-FourierGrid models follow cfg_train.ray_sampler
    cfg.model = "DVGO"
    out = tr.gather_training_rays(types.SimpleNamespace(), {"irregular_shape": True}, imgs, cfg, [0, 1, 2], cfg_train, poses, HW, Ks, {}, device="cpu")
    check(out[1], g["flat_o"], True)
    assert out[4] is None
