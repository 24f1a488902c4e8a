"""Config surface of reference `utils/arg_helper.py:8-76` without the missing third-party
`easydict`: YAML -> attribute dictionary, same derived keys."""
import argparse
import os
import time

import yaml


class AttrDict(dict):
    """EasyDict stand-in: nested attribute access; missing keys raise AttributeError so the
    reference's `hasattr(config.model, 'dropout')` probing (model/lanczos_net.py:24) works."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


def parse_arguments(argv=None):
    p = argparse.ArgumentParser(description='Running Experiments of LanczosNet (MI355X path)')
    p.add_argument('-c', '--config_file', type=str, default='config/qm8_lanczos_net.yaml',
                   required=True, help='Path of config file')
    p.add_argument('-l', '--log_level', type=str, default='INFO',
                   help='Logging Level, DEBUG, INFO, WARNING, ERROR, CRITICAL')
    p.add_argument('-m', '--comment', help='Experiment comment')
    p.add_argument('-t', '--test', help='Test model', action='store_true')
    return p.parse_args(argv)


def load_config(config_file):
    with open(config_file, 'r') as f:
        return AttrDict(yaml.safe_load(f))


def get_config(config_file, exp_dir=None):
    """utils/arg_helper.py:36-67: derive run_id / exp_name / save_dir, snapshot the config."""
    config = load_config(config_file)
    config.run_id = str(os.getpid())
    config.exp_name = '_'.join([config.model.name, config.dataset.name,
                                time.strftime('%Y-%b-%d-%H-%M-%S'), config.run_id])
    if exp_dir is not None:
        config.exp_dir = exp_dir
    config.save_dir = os.path.join(config.exp_dir, config.exp_name)
    os.makedirs(config.save_dir, exist_ok=True)
    with open(os.path.join(config.save_dir, 'config.yaml'), 'w') as f:
        yaml.safe_dump(_plain(config), f, default_flow_style=False)
    return config


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


def make_model_config(cfg, name='LanczosNet', general=False, loss='MSE'):
    """Build the `config` object a model constructor reads from a flat dict (tests, bench)."""
    model = dict(name=name, short_diffusion_dist=list(cfg['short_diffusion_dist']),
                 long_diffusion_dist=list(cfg['long_diffusion_dist']),
                 num_eig_vec=cfg['num_eig_vec'],
                 spectral_filter_kind=cfg['spectral_filter_kind'], input_dim=cfg['input_dim'],
                 hidden_dim=list(cfg['hidden_dim']), output_dim=cfg['output_dim'],
                 num_layer=cfg['num_layer'], loss=loss)
    if general:
        dataset = dict(node_emb_dim=cfg['input_dim'], graph_emb_dim=cfg['output_dim'],
                       num_edge_type=cfg['num_bond_type'])
    else:
        dataset = dict(num_atom=cfg['num_atom'], num_bond_type=cfg['num_bond_type'])
    return AttrDict(dict(seed=1234, dataset=dataset, model=model))
