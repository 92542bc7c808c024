"""GPU: the MC-dropout prediction of a scan stream replayed as one HIP graph (semanticlidarunc_amd/graph_infer.py) equals the eager
`utils.mc_dropout.mc_predict` with the same generator state -- bit for bit (the same kernels on the same multipliers) -- draws fresh masks on
every call, and follows the input buffer."""
import pytest
import torch

from semanticlidarunc_amd import salsanext as sn
from semanticlidarunc_amd.graph_infer import GraphedMCPredict
from semanticlidarunc_amd.salsanext import SalsaNext
from semanticlidarunc_amd.testing import seeded_model, synthetic_scan
from semanticlidarunc_amd.utils.mc_dropout import mc_predict

pytestmark = pytest.mark.gpu


def test_graphed_stream_equals_eager_mc_predict(cuda):
    model = seeded_model(SalsaNext).to(cuda).eval()
    xa, _ = synthetic_scan(1, 32, 256, seed=5)
    xb, _ = synthetic_scan(1, 32, 256, seed=6)
    xa, xb = xa.to(cuda), xb.to(cuda)
    sn.set_conv_precision("f16")
    try:
        stream = GraphedMCPredict(model, xa, T=4)
        for x in (xa, xb):
            torch.manual_seed(123)
            with torch.no_grad():
                want = [t.clone() for t in mc_predict(model, [x], T=4)]
            torch.manual_seed(123)
            got = [t.clone() for t in stream(x)]
            for w, g in zip(want, got):
                assert torch.equal(w, g)
        first = [t.clone() for t in stream(xa)]
        second = [t.clone() for t in stream(xa)]
        assert not torch.equal(first[0], second[0])                 # the generator advanced: fresh masks per call
        with pytest.raises(RuntimeError):
            stream(torch.zeros(2, 5, 32, 256, device=cuda))
    finally:
        sn.set_conv_precision("fp32")
    with pytest.raises(RuntimeError):
        GraphedMCPredict(model, xa, T=4)                             # fp32 precision: not the fused half-precision path
