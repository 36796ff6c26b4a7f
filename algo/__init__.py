"""Import-path shim: ``algo`` IS ``isaacgyminsertion_amd.algo``.

The reference's entry points import the trainers by their in-tree path (isaacgyminsertion/train.py:31-32,
``from algo.ppo.frozen_ppo import PPO`` / ``from algo.ext_adapt.ext_adapt import ExtrinsicAdapt``;
train_supervised.py:40, ``from algo.models.transformer.runner import Runner``; algo/deploy/deploy_s{1,2}.py).  With
this repository's root ahead of the reference's on ``PYTHONPATH`` those statements resolve here, unchanged: every
``algo.*`` module name is registered as an alias of the SAME module object as ``isaacgyminsertion_amd.algo.*`` (one
class object per class, whichever way it was imported)."""
import importlib
import pkgutil
import sys

import isaacgyminsertion_amd.algo as _real

for _m in pkgutil.walk_packages(_real.__path__, prefix="isaacgyminsertion_amd.algo."):
    _mod = importlib.import_module(_m.name)
    sys.modules["algo" + _m.name[len("isaacgyminsertion_amd.algo"):]] = _mod
sys.modules["algo"] = _real
