"""Drop-in for ``KL_offClasses_to_uniform`` of the reference (``src/losses/regularizers.py:291-389``, the ``w_kl`` term of the
default Dirichlet loss): KL(Dir(alpha~) || Dir(1, ..., 1)) with the true class's alpha replaced by 1, mean over valid pixels,
as one fused HIP forward / backward pass (``csrc/dirichlet_loss.hip``).  The confidence-weighted variant
(``with_conf_weighting=True``, not used by the reference's Trainer) and the other regularizers are not mirrored."""
from __future__ import annotations

from typing import Optional

import torch.nn as nn

from .dirichlet_losses import _check_ignore, _DirichletLossFn


class KL_offClasses_to_uniform(nn.Module):
    def __init__(self, ignore_index: Optional[int] = None, with_conf_weighting: bool = False, gamma: float = 1.0, eps: float = 1e-8):
        super().__init__()
        if with_conf_weighting:
            raise NotImplementedError("with_conf_weighting=True is not implemented on the HIP path")
        self.ignore_index, self.eps, self.with_conf_weighting, self.gamma = _check_ignore(ignore_index), eps, False, gamma

    def forward(self, alpha, target):
        return _DirichletLossFn.apply(alpha, target, "kl_off_uniform", 0.0, self.eps, self.ignore_index)


# drop-in mode (this file shadows the reference's module of the same import path): names it does not define come from there
from semanticlidarunc_amd._shadow import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(__name__, __file__, globals())
