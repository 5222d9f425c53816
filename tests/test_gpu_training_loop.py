"""End-to-end training on the GPU: the loss falls and held-out PSNR rises when the
hash-grid / frequency models are fitted to an analytic teacher field through the full
stage chain (trace -> sample -> encode -> MLP fwd -> composite -> L2 -> composite bwd ->
MLP bwd -> hash scatter -> Adam)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("encoding,steps,gain", [("hash", 300, 7.0), ("freq", 300, 5.0)])
def test_training_converges(gpu, encoding, steps, gain):
    import train_demo
    p0, p1, losses = train_demo.run(steps=steps, encoding=encoding, grid=32, res=64, batch=4096, n_poses=12, verbose=False)
    assert losses[-1] < 0.1 * losses[0], losses
    assert p1 > p0 + gain, (p0, p1)          # held-out pose, measured: hash +10.5 dB, freq +7.7 dB
