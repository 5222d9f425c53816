"""End-to-end training on the GPU: the loss falls and held-out PSNR rises when the
hash-grid / frequency models are fitted to an analytic teacher field through the full
stage chain (trace -> sample -> encode -> MLP fwd -> composite -> L2 -> composite bwd ->
MLP bwd -> hash scatter -> Adam)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("encoding,steps,gain", [("hash", 250, 8.0), ("freq", 250, 4.0)])
def test_training_converges(gpu, encoding, steps, gain):
    import train_demo
    p0, p1, losses = train_demo.run(steps=steps, encoding=encoding, grid=32, res=48, batch=4096, n_poses=8, verbose=False)
    assert losses[-1] < 0.5 * losses[0], losses
    assert p1 > p0 + gain, (p0, p1)
