"""Device-resident sample columns with the capped-buffer policy of the reference's aggregators
(``metrics/auroc.py:125-141``, ``models/evaluator.py:681-700``): unlimited append when ``cap`` is None; otherwise fill up to
``cap`` (a random subset of the incoming batch when it does not fit), and once full keep each incoming sample with
probability ``cap / seen`` and let the kept ones overwrite random slots.  The random choices come from a numpy Generator on
the host -- the same calls, in the same order, as the reference makes -- and are applied to tensors that stay on the GPU."""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch


class CappedColumns:
    def __init__(self, cap: Optional[int], seed: int = 0):
        self.cap = cap
        self.rng = np.random.default_rng(seed)
        self.clear()

    def clear(self) -> None:
        self.columns: Optional[List[torch.Tensor]] = None
        self.seen = 0

    def __len__(self) -> int:
        return 0 if self.columns is None else int(self.columns[0].numel())

    def _extend(self, cols: Sequence[torch.Tensor]) -> None:
        self.columns = list(cols) if self.columns is None else [torch.cat([old, new]) for old, new in zip(self.columns, cols)]

    def push(self, *cols: torch.Tensor) -> None:
        """Add one batch of equally long 1-D columns."""
        incoming = int(cols[0].numel())
        if incoming == 0:
            return
        self.seen += incoming
        if self.cap is None:
            self._extend(cols)
            return
        dev = cols[0].device
        room = self.cap - len(self)
        if room > 0:
            if incoming > room:                                    # keep a uniformly drawn subset that fits
                pick = torch.from_numpy(self.rng.choice(incoming, size=room, replace=False)).to(dev)
                cols = [c[pick] for c in cols]
            self._extend(cols)
            return
        accept = self.rng.random(incoming) < min(1.0, float(self.cap) / float(self.seen + 1e-9))
        if accept.any():
            accept_t = torch.from_numpy(accept).to(dev)
            cols = [c[accept_t] for c in cols]
            slots = torch.from_numpy(self.rng.choice(self.cap, size=int(cols[0].numel()), replace=False)).to(dev)
            for buf, c in zip(self.columns, cols):
                buf[slots] = c
