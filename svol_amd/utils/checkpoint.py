"""Checkpoint I/O in the reference's schema (train.py:145-154, 267-284; test.py:72-88) — SURVEY.md §8 f4, host side.

A reference ``.ckpt`` is ``torch.save`` of ``{'model', 'optimizer', 'lr_scheduler', 'amp', 'iter', 'args'}``; under apex DDP
every model key carries a ``module.`` prefix, which test.py strips by hand.  The modules of this build keep the reference's
state-dict keys, so loading is key surgery only:

* ``module.`` prefix stripped (test.py:76-86 looks at the first key; here every key is checked);
* ``backbone.*`` entries of a checkpoint trained with a torchvision ResNet are set aside when the model was built with a
  parameter-free backbone (``--backbone features``) and returned to the caller instead of failing ``load_state_dict``;
* ``amp``: apex loss-scaler state.  bf16 / fp32 training needs no loss scaling; a neutral ``loss_scaler0`` entry is written so
  that a reference driver's ``amp.load_state_dict(checkpoint['amp'])`` (train.py:151) accepts the file, and whatever a
  reference file holds is passed back untouched.
"""
from __future__ import annotations

import argparse
import os
from collections import OrderedDict
from types import SimpleNamespace
from typing import Optional

import torch

NEUTRAL_AMP_STATE = {'loss_scaler0': {'loss_scale': 1.0, 'unskipped': 0}}


def checkpoint_name(args, iter_i: int) -> str:
    """file name pattern of train.py:279-283"""
    return (f'{iter_i:04d}_model_{args.video_dataset}_{args.sketch_dataset}_{args.sketch_head}_{args.backbone}_'
            f'{args.num_layers}l_{args.num_frames}f_{args.num_queries}q_'
            f'{args.set_cost_bbox}_{args.set_cost_giou}_{args.set_cost_class}.ckpt')


def strip_module_prefix(state_dict):
    """test.py:76-86 (there: 7 characters off every key when the first key contains 'module')."""
    return OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in state_dict.items())


def save_checkpoint(path, model, optimizer, lr_scheduler, iter_i: int, args, amp_state=None, ddp_prefix: bool = False):
    """train.py:267-284.  ``ddp_prefix=True`` writes the ``module.``-prefixed keys an apex-DDP run would."""
    sd = model.state_dict()
    if ddp_prefix:
        sd = OrderedDict(('module.' + k, v) for k, v in sd.items())
    if isinstance(args, SimpleNamespace):  # keep the file loadable by torch.load(weights_only=True) + argparse.Namespace
        args = argparse.Namespace(**vars(args))
    ckpt = {'model': sd, 'optimizer': optimizer.state_dict() if optimizer is not None else None,
            'lr_scheduler': lr_scheduler.state_dict() if lr_scheduler is not None else None,
            'amp': amp_state if amp_state is not None else NEUTRAL_AMP_STATE, 'iter': int(iter_i), 'args': args}
    # not in the reference's schema (its drivers ignore unknown entries): where the stateless dropout masks of the enc/dec Transformer
    # continue after a resume — module path -> {'drop_base_seed', 'drop_step'}.  (amp_state: pass DynamicLossScaler.state_dict() for fp16.)
    rng = {n: m.dropout_state() for n, m in model.named_modules() if hasattr(m, 'dropout_state')}
    if rng:
        ckpt['svol_dropout'] = rng
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ckpt, path)
    return path


def load_checkpoint(path, model, optimizer=None, lr_scheduler=None, resume_all: bool = False, map_location='cpu',
                    trust_pickle: bool = False):
    """train.py:145-154 / test.py:72-88.  Returns ``(checkpoint, info)``; ``info['start_iter']`` is set with ``resume_all``,
    ``info['set_aside']`` lists checkpoint entries the model has no slot for (a ResNet backbone's weights when the model
    takes pre-extracted features).  ``trust_pickle``: the reference pickles its argparse namespace into the file; by default
    only tensors, containers and ``argparse.Namespace`` are unpickled."""
    if trust_pickle:
        ckpt = torch.load(path, map_location=map_location, weights_only=False)
    else:
        with torch.serialization.safe_globals([argparse.Namespace]):
            ckpt = torch.load(path, map_location=map_location, weights_only=True)
    sd = strip_module_prefix(ckpt['model'])
    own = model.state_dict()
    set_aside = []
    if not any(k.startswith('backbone.') for k in own):
        set_aside = [k for k in sd if k.startswith('backbone.')]
        for k in set_aside:
            sd.pop(k)
    model.load_state_dict(sd)  # strict: a missing or unexpected head key is an error, as in the reference
    info = {'iter': ckpt.get('iter'), 'set_aside': set_aside, 'amp': ckpt.get('amp')}
    if resume_all and ckpt.get('svol_dropout'):
        mods = dict(model.named_modules())
        for n, st in ckpt['svol_dropout'].items():
            if n in mods and hasattr(mods[n], 'load_dropout_state'):
                mods[n].load_dropout_state(st)
    if resume_all:
        if optimizer is not None and ckpt.get('optimizer') is not None:
            optimizer.load_state_dict(ckpt['optimizer'])
        if lr_scheduler is not None and ckpt.get('lr_scheduler') is not None:
            lr_scheduler.load_state_dict(ckpt['lr_scheduler'])
        info['start_iter'] = ckpt['iter'] + 1
    return ckpt, info
