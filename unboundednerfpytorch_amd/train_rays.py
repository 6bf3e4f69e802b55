"""Ray-batch side of the training loop (run_train.py:129-147,203-247).

The reference prepares ALL training rays once -- `FourierGridModel.gather_training_rays`
(FourierGrid_model.py:298-333) -> `FourierGrid_get_training_rays` (:264-296) / `dvgo.get_training_rays_flatten`
(dvgo.py:594-616) / `dvgo.get_training_rays` (:562-590) / `dvgo.get_training_rays_in_maskcache_sampling` (:619-657) -- and
draws N_rand of them per iteration with `batch_indices_generator` (dvgo.py:660-668) or `torch.randint`
(run_train.py:203-236), copying the batch to the GPU when `load2gpu_on_the_fly` (run_train.py:238-245).

Same functions, same return tuples and the same random streams here (numpy permutation for the sampler, torch.randint for
the 'random' mode), with the rays produced by the one-kernel `fourier_render.get_rays_of_a_view` when the poses live on the
GPU.  On an MI355X the natural set-up is `load2gpu_on_the_fly = False`: 250 full-HD views are 500 M rays x 48 B = 25 GB of
the 288 GB, so the ray table stays resident and a batch is four device-side row gathers -- no per-iteration H2D copy.
NDC rays belong to the DirectMPIGO path (out of scope)."""
import numpy as np
import torch

from .fourier_render import get_rays_of_a_view

FOURIERGRID_DATASETS = ("waymo", "mega", "nerfpp")      # FourierGrid_model.py:307


def _rays(H, W, K, c2w, ndc, inverse_y, flip_x, flip_y, device):
    if ndc:
        raise NotImplementedError("NDC rays belong to the DirectMPIGO path (out of scope, SURVEY.md section 2)")
    c2w = torch.as_tensor(c2w, dtype=torch.float32)
    if device.type == "cuda":
        c2w = c2w.to(device)         # device-resident pose: ONE kernel per view (ugrid_rays_of_a_view)
    o, d, v = get_rays_of_a_view(int(H), int(W), K, c2w, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y)
    return o.to(device), d.to(device), v.to(device)


@torch.no_grad()
def get_training_rays(rgb_tr, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y):
    """dvgo.get_training_rays (dvgo.py:562-590): equal-sized views, rays kept as [N,H,W,3]."""
    assert len(np.unique(np.asarray(HW), axis=0)) == 1
    assert len(np.unique(np.asarray(Ks).reshape(len(Ks), -1), axis=0)) == 1
    assert len(rgb_tr) == len(train_poses) and len(rgb_tr) == len(Ks) and len(rgb_tr) == len(HW)
    H, W = (int(x) for x in HW[0])
    K = Ks[0]
    dev = rgb_tr.device
    rays_o_tr = torch.zeros([len(rgb_tr), H, W, 3], device=dev)
    rays_d_tr, viewdirs_tr = torch.zeros_like(rays_o_tr), torch.zeros_like(rays_o_tr)
    for i, c2w in enumerate(train_poses):
        o, d, v = _rays(H, W, K, c2w, ndc, inverse_y, flip_x, flip_y, dev)
        rays_o_tr[i].copy_(o)
        rays_d_tr[i].copy_(d)
        viewdirs_tr[i].copy_(v)
    return rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, [1] * len(rgb_tr)


def _flatten(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y, with_index):
    assert len(rgb_tr_ori) == len(train_poses) and len(rgb_tr_ori) == len(Ks) and len(rgb_tr_ori) == len(HW)
    dev = rgb_tr_ori[0].device
    total = sum(im.shape[0] * im.shape[1] for im in rgb_tr_ori)
    rgb_tr = torch.zeros([total, 3], device=dev)
    rays_o_tr, rays_d_tr, viewdirs_tr = torch.zeros_like(rgb_tr), torch.zeros_like(rgb_tr), torch.zeros_like(rgb_tr)
    indexs_tr = torch.zeros_like(rgb_tr) if with_index else None       # image index, float [N,3] like the reference
    imsz, top = [], 0
    for cur, (c2w, img, (H, W), K) in enumerate(zip(train_poses, rgb_tr_ori, HW, Ks)):
        H, W = int(H), int(W)
        assert tuple(img.shape[:2]) == (H, W)
        o, d, v = _rays(H, W, K, c2w, ndc, inverse_y, flip_x, flip_y, dev)
        n = H * W
        rgb_tr[top:top + n].copy_(img.flatten(0, 1))
        rays_o_tr[top:top + n].copy_(o.flatten(0, 1))
        rays_d_tr[top:top + n].copy_(d.flatten(0, 1))
        viewdirs_tr[top:top + n].copy_(v.flatten(0, 1))
        if with_index:
            indexs_tr[top:top + n] = float(cur)
        imsz.append(n)
        top += n
    assert top == total
    return rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, indexs_tr, imsz


@torch.no_grad()
def get_training_rays_flatten(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y):
    """dvgo.get_training_rays_flatten (dvgo.py:594-616): views of any size, all pixels, flattened to [N,3]."""
    rgb, o, d, v, _, imsz = _flatten(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y, False)
    return rgb, o, d, v, imsz


@torch.no_grad()
def FourierGrid_get_training_rays(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y, pos_emb=None):
    """FourierGridModel.FourierGrid_get_training_rays (FourierGrid_model.py:264-296): the flattened rays plus the image
    index of every ray; pos_emb (the model's optional pose refinement) is added to the camera centres IN PLACE, as the
    reference does."""
    if pos_emb is not None:
        train_poses[:, :3, 3] = train_poses[:, :3, 3] + pos_emb
    return _flatten(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y, True)


def batch_indices_generator(N, BS):
    """dvgo.batch_indices_generator (dvgo.py:660-668): epochs of a numpy permutation cut into BS-sized index batches; an
    incomplete tail is dropped and a new permutation drawn.  Same numpy random stream as the reference."""
    idx, top = torch.LongTensor(np.random.permutation(N)), 0
    while True:
        if top + BS > N:
            idx, top = torch.LongTensor(np.random.permutation(N)), 0
        yield idx[top:top + BS]
        top += BS


def _get(cfg, name, default=None):
    return cfg.get(name, default) if isinstance(cfg, dict) else getattr(cfg, name, default)


def gather_training_rays(model, data_dict, images, cfg, i_train, cfg_train, poses, HW, Ks, render_kwargs, device=None):
    """FourierGridModel.gather_training_rays (FourierGrid_model.py:298-333).  cfg / cfg.data / cfg_train: attribute objects
    or dicts.  Returns (rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, indexs_train, imsz, batch_index_sampler)."""
    data = _get(cfg, 'data')
    if device is None:
        device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
    store = torch.device('cpu') if _get(data, 'load2gpu_on_the_fly', False) else torch.device(device)
    if _get(data_dict, 'irregular_shape'):
        rgb_tr_ori = [images[i].to(store) for i in i_train]
    else:
        rgb_tr_ori = images[i_train].to(store)
    kw = dict(train_poses=poses[i_train], HW=HW[i_train], Ks=Ks[i_train], ndc=_get(data, 'ndc', False),
              inverse_y=_get(data, 'inverse_y', False), flip_x=_get(data, 'flip_x', False), flip_y=_get(data, 'flip_y', False))
    indexs_train = None
    sampler = _get(cfg_train, 'ray_sampler')
    if _get(data, 'dataset_type') in FOURIERGRID_DATASETS or _get(cfg, 'model') == 'FourierGrid':
        rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, indexs_train, imsz = FourierGrid_get_training_rays(
            rgb_tr_ori=rgb_tr_ori, pos_emb=getattr(model, 'pos_emb', None), **kw)
    elif sampler == 'in_maskcache':
        from .dvgo_render import get_training_rays_in_maskcache_sampling
        rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, imsz = get_training_rays_in_maskcache_sampling(
            rgb_tr_ori=rgb_tr_ori, model=model, render_kwargs=render_kwargs, **kw)
    elif sampler == 'flatten':
        rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, imsz = get_training_rays_flatten(rgb_tr_ori=rgb_tr_ori, **kw)
    else:
        rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, imsz = get_training_rays(rgb_tr=rgb_tr_ori, **kw)
    index_generator = batch_indices_generator(len(rgb_tr), _get(cfg_train, 'N_rand'))
    return rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, indexs_train, imsz, (lambda: next(index_generator))


def sample_batch(cfg_train, rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, indexs_tr, batch_index_sampler, device=None,
                 load2gpu_on_the_fly=False):
    """One iteration's ray batch (run_train.py:203-245): returns (target, rays_o, rays_d, viewdirs, indexs).
    'flatten' / 'in_maskcache': rows chosen by batch_index_sampler(); 'random': torch.randint over the leading dims of
    the (image-shaped or flattened) ray table, on its device -- the same draws in the same order as the reference.
    With load2gpu_on_the_fly the five tensors are copied to `device`; with a resident table (the MI355X set-up) the index
    tensor goes to the table's device and the batch is four row gathers there."""
    sampler = _get(cfg_train, 'ray_sampler')
    n_rand = _get(cfg_train, 'N_rand')
    if sampler in ('flatten', 'in_maskcache'):
        sel = batch_index_sampler()
        if sel.device != rgb_tr.device:
            sel = sel.to(rgb_tr.device, non_blocking=True)
        sel = (sel,)
    elif sampler == 'random':
        if rgb_tr.dim() != 2:
            sel = tuple(torch.randint(rgb_tr.shape[k], [n_rand], device=rgb_tr.device) for k in range(3))
        else:
            sel_b = torch.randint(rgb_tr.shape[0], [n_rand], device=rgb_tr.device)
            torch.randint(rgb_tr.shape[1], [n_rand], device=rgb_tr.device)       # the reference draws (and ignores) sel_r
            sel = (sel_b,)
    else:
        raise NotImplementedError(sampler)
    out = [t[sel] if t is not None else None for t in (rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, indexs_tr)]
    if load2gpu_on_the_fly:
        out = [t.to(device) if t is not None else None for t in out]
    return tuple(out)
