"""
srvp_amd -- MI355X-native (gfx950 / CDNA4) implementation of the SRVP training / rollout hot path.

Public surface mirrors the reference (edouardelasalles/srvp): `StochasticLatentResidualVideoPredictor`
(module/srvp.py), `train.train / evaluate / main` (train.py), `args.create_args` (args.py), `helper.DotDict`.
All arithmetic runs in libsrvp_hip.so (srvp_amd/csrc, C ABI in include/srvp_hip.h).
"""
from .model import StochasticLatentResidualVideoPredictor  # noqa: F401
from .optim import FusedAdam  # noqa: F401
from .helper import DotDict  # noqa: F401

__all__ = ['StochasticLatentResidualVideoPredictor', 'FusedAdam', 'DotDict']
