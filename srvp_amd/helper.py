"""Configuration helpers with the reference's names (reference helper.py:20-44)."""
import json

import yaml


class DotDict(dict):
    """dict with attribute access; a missing key reads as None (the reference relies on that for optional flags)."""

    def __getattr__(self, k):
        return self.get(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]


def load_yaml(path):
    with open(path, 'r') as f:
        return DotDict(yaml.safe_load(f))


def load_json(path):
    with open(path, 'r') as f:
        return DotDict(json.load(f))
