"""Offline supervised student training entry point (isaacgyminsertion/train_supervised.py:40-45):

    python -m isaacgyminsertion_amd.train_supervised [--config my.yaml] \\
        offline_train.data_folder=/data/tactile_insertion offline_train.train.epochs=10

builds ``Runner(cfg, agent=None)`` and calls ``run()``.  The reference composes its config with Hydra; here
a plain YAML file (same keys) is merged over the built-in defaults and ``a.b.c=value`` overrides are applied.
"""
import argparse

import yaml

from .algo.models.transformer.runner import Runner
from .algo.models.transformer.utils import set_seed
from .utils.config import default_config, load_config, merge


def _override(cfg, dotted, value):
    node = {}
    cur = node
    keys = dotted.split('.')
    for k in keys[:-1]:
        cur[k] = {}
        cur = cur[k]
    cur[keys[-1]] = yaml.safe_load(value)
    return merge(cfg, node)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--config', default=None, help='YAML file with the reference config keys')
    ap.add_argument('overrides', nargs='*', help='a.b.c=value')
    args = ap.parse_args(argv)
    cfg = load_config(args.config) if args.config else default_config()
    cfg = merge(cfg, {'offline_training': True})
    for o in args.overrides:
        k, _, v = o.partition('=')
        cfg = _override(cfg, k, v)
    set_seed(cfg.offline_train.get('seed', 0))
    runner = Runner(cfg, agent=None)
    runner.run()
    return runner


if __name__ == "__main__":
    main()
