"""Helpers to read the fixtures written by tests/make_golden.py."""
import ast
import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
OUT_NAMES = ['x_', 'y', 'z', 'w', 'q_y_0_params', 'q_z_params', 'p_z_params', 'res']
CFG_KEYS = ['nx', 'nc', 'nf', 'nhx', 'ny', 'nz', 'skipco', 'nt_inf', 'nh_inf', 'nlayers_inf', 'nh_res', 'nlayers_res',
            'archi']


def fixture_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, '*.npz'))
                  if not os.path.basename(p).startswith(('known', 'metrics', 'mmnist', 'full_', 'dense_')))


def full_fixture_names():
    """Full-width reference runs of the BASELINE.json shapes (tests/make_golden.py gen_full)."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'full_*.npz')))


def dense_fixture_names():
    """remove_intermediate=False eval forwards of the reference (tests/make_golden.py gen_dense)."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'dense_*.npz')))


class Fixture:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
        self.meta = ast.literal_eval(str(self.z['meta']))
        self.cfg = dict(zip(CFG_KEYS, self.meta['ctor']))

    def t(self, key, dtype=None):
        a = torch.from_numpy(np.array(self.z[key]))
        return a.to(dtype) if dtype is not None and a.is_floating_point() else a

    def has(self, key):
        return key in self.z.files

    def group(self, prefix, dtype=None):
        return {k[len(prefix):]: self.t(k, dtype) for k in self.z.files if k.startswith(prefix)}

    def state(self, which='sd0', dtype=None):
        return self.group(which + '.', dtype)

    def tape(self, prefix='tape.', dtype=None):
        return self.group(prefix, dtype)
