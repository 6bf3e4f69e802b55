"""Optimizer factory and checkpoint I/O of the training program, on this package's model / optimizers:
  create_optimizer_or_freeze_model  <- FourierGrid/utils.py:26-56 (MaskedAdam param groups from the `lrate_*` keys of the
                                       train config, exponential decay by global_step, freezing at lr <= 0)
  save_checkpoint / load_checkpoint / load_model
                                    <- FourierGrid_ckpt_manager.py:44-60, utils.py:61-74: the reference's .tar layout
                                       {global_step, model_kwargs, model_state_dict, optimizer_state_dict}, so
                                       checkpoints written by either side load on the other."""
import torch
import torch.nn as nn

from .masked_adam import MaskedAdam
from .sharded_adam import ShardedMaskedAdam


def _cfg_get(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


def create_optimizer_or_freeze_model(model, cfg_train, global_step, verbose=False, sharded=False, group=None, ops=None,
                                     recycle_grads=True):
    """cfg_train: mapping or attribute object with `lrate_<field>` entries, `lrate_decay` (in thousands of steps) and
    `skip_zero_grad_fields`.  Every `lrate_<field>` whose <field> is an attribute of the model becomes one param group
    with lr = lrate * 0.1 ** (global_step / (lrate_decay * 1000)); lr <= 0 freezes the field instead.
    sharded=True builds a ShardedMaskedAdam over `group` (data-parallel training), otherwise a MaskedAdam.
    recycle_grads (this package's training loop, train_step.train_iteration: on): the grid gradients' buffers are
    re-zeroed by the update kernels and recycled, `.grad` is None after step() -- pass False for code that reads gradients
    after the step (the drop-in MaskedAdam class itself defaults to the reference's behaviour)."""
    keys = list(cfg_train.keys())
    decay = 0.1 ** (global_step / (_cfg_get(cfg_train, 'lrate_decay') * 1000))
    skip_fields = _cfg_get(cfg_train, 'skip_zero_grad_fields')
    groups = []
    for key in keys:
        if not key.startswith('lrate_'):
            continue
        field = key[len('lrate_'):]
        target = getattr(model, field, None) if hasattr(model, field) else None
        if target is None:
            continue
        lr = _cfg_get(cfg_train, key) * decay
        if lr > 0:
            params = target.parameters() if isinstance(target, nn.Module) else target
            groups.append({'params': params, 'lr': lr, 'skip_zero_grad': field in skip_fields})
            if verbose:
                print('create_optimizer_or_freeze_model: %s lr %g' % (field, lr))
        else:
            if isinstance(target, nn.Module):
                for p in target.parameters():
                    p.requires_grad = False
            else:
                target.requires_grad = False
    if sharded:
        return ShardedMaskedAdam(groups, group=group, ops=ops, recycle_grads=recycle_grads)
    opt = MaskedAdam(groups, recycle_grads=recycle_grads and ops is None)
    if ops is not None:
        opt.ops = ops
    return opt


def _canonical(obj):
    """state-dict tensors in the reference's row-major layout (the channel-last training layout of multi-channel grids is
    a storage detail of this package; files stay byte-compatible with the reference's)"""
    if torch.is_tensor(obj):
        return obj.contiguous()
    if isinstance(obj, dict):
        out = type(obj)((k, _canonical(v)) for k, v in obj.items())
        if hasattr(obj, '_metadata'):            # nn.Module.state_dict()'s version metadata (load_state_dict reads it)
            out._metadata = obj._metadata
        return out
    if isinstance(obj, (list, tuple)):
        return type(obj)(_canonical(v) for v in obj)
    return obj


def save_checkpoint(path, model, optimizer, global_step):
    torch.save({'global_step': global_step, 'model_kwargs': model.get_kwargs(), 'model_state_dict': _canonical(model.state_dict()),
                'optimizer_state_dict': _canonical(optimizer.state_dict())}, path)


def load_model(ckpt_path, model_class=None, **model_extra):
    """-> (model, model_kwargs); model_class defaults to fourier_model.FourierGridModel; model_extra is passed on to the
    constructor (e.g. backend=... in tests)."""
    if model_class is None:
        from .fourier_model import FourierGridModel as model_class
    ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)
    kwargs = dict(ckpt['model_kwargs'])
    model = model_class(**kwargs, **model_extra)
    model.load_state_dict(ckpt['model_state_dict'])
    return model, kwargs


def load_checkpoint(model, optimizer, ckpt_path, no_reload_optimizer):
    ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)
    model.load_state_dict(ckpt['model_state_dict'])
    if not no_reload_optimizer:
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
    return model, optimizer, ckpt['global_step']
