"""Golden vectors for the ray-batch side of the training loop (unboundednerfpytorch_amd/train_rays.py), produced by the
REFERENCE's own functions imported from /root/reference in the build container:
dvgo.get_training_rays / get_training_rays_flatten / batch_indices_generator (dvgo.py:562-616,660-668),
FourierGridModel.FourierGrid_get_training_rays (FourierGrid_model.py:264-296) and the per-iteration selection of
run_train.py:203-236 re-executed line by line.  Run:  python tests/golden/gen_train_rays_golden.py  -> train_rays.npz"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import install_stubs  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_rays.npz")


def scene(seed=3):
    g = torch.Generator().manual_seed(seed)
    sizes = [(6, 9), (5, 7), (6, 9)]
    imgs = [torch.rand(h, w, 3, generator=g) for h, w in sizes]
    poses = torch.eye(4)[:3].repeat(3, 1, 1) + 0.3 * torch.randn(3, 3, 4, generator=g)
    Ks = np.array([[[40.0 + 3 * i, 0, w / 2.0], [0, 41.0 + 2 * i, h / 2.0], [0, 0, 1]] for i, (h, w) in enumerate(sizes)], dtype=np.float32)
    HW = np.array(sizes)
    return imgs, poses, HW, Ks


def main():
    dvgo = install_stubs.import_reference("dvgo")
    fgm = install_stubs.import_reference("FourierGrid_model")
    imgs, poses, HW, Ks = scene()
    out = {}
    flags = dict(ndc=False, inverse_y=True, flip_x=False, flip_y=True)
    r = dvgo.get_training_rays_flatten(rgb_tr_ori=imgs, train_poses=poses, HW=HW, Ks=Ks, **flags)
    for k, v in zip(("rgb", "o", "d", "v"), r[:4]):
        out["flat_" + k] = v.numpy()
    out["flat_imsz"] = np.array(r[4])
    self = types.SimpleNamespace(pos_emb=None)
    r = fgm.FourierGridModel.FourierGrid_get_training_rays(self, rgb_tr_ori=imgs, train_poses=poses.clone(), HW=HW, Ks=Ks, **flags)
    for k, v in zip(("rgb", "o", "d", "v", "idx"), r[:5]):
        out["fg_" + k] = v.numpy()
    out["fg_imsz"] = np.array(r[5])
    self = types.SimpleNamespace(pos_emb=torch.tensor([0.1, -0.2, 0.05]))
    p2 = poses.clone()
    r = fgm.FourierGridModel.FourierGrid_get_training_rays(self, rgb_tr_ori=imgs, train_poses=p2, HW=HW, Ks=Ks, **flags)
    out["fgpos_o"] = r[1].numpy()
    out["fgpos_poses_after"] = p2.numpy()
    # equal-sized views, image-shaped table
    same = torch.stack([imgs[0], imgs[2]])
    r = dvgo.get_training_rays(rgb_tr=same, train_poses=poses[[0, 2]], HW=HW[[0, 2]], Ks=np.stack([Ks[0], Ks[0]]),
                               ndc=False, inverse_y=False, flip_x=True, flip_y=False)
    for k, v in zip(("rgb", "o", "d", "v"), r[:4]):
        out["img_" + k] = v.numpy()
    # the sampler: numpy stream
    np.random.seed(11)
    gen = dvgo.batch_indices_generator(50, 16)
    out["sampler_batches"] = np.stack([next(gen).numpy() for _ in range(7)])
    # run_train.py:203-236, 'random' mode on both table shapes (torch stream)
    torch.manual_seed(5)
    rgb_tr, o_tr = r[0], r[1]
    n_rand = 10
    sel_b = torch.randint(rgb_tr.shape[0], [n_rand]); sel_r = torch.randint(rgb_tr.shape[1], [n_rand]); sel_c = torch.randint(rgb_tr.shape[2], [n_rand])
    out["rand3_target"] = rgb_tr[sel_b, sel_r, sel_c].numpy()
    out["rand3_o"] = o_tr[sel_b, sel_r, sel_c].numpy()
    flat_rgb = torch.from_numpy(out["flat_rgb"])
    sel_b = torch.randint(flat_rgb.shape[0], [n_rand]); torch.randint(flat_rgb.shape[1], [n_rand])
    out["rand2_target"] = flat_rgb[sel_b].numpy()
    out["rand_next"] = torch.rand(3).numpy()       # the stream position after both draws
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
