"""The oracle's restatement of the SuperSloMo interpolation (oracle/interp_ref.py) against G10 — outputs of the
reference's own UNet / backWarp / interpolate.forward (tools/gen_golden.py gen_g10) — and the product's weight container
(avtex.slowmo.UNet) against the same vectors: the state-dict layout and the architecture are the reference's."""
import os

import numpy as np
import pytest
import torch

from interp_weights import frame_pair, unet_state
from oracle import interp_ref

G10 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g10_interp.npz"))


def _case(name):
    h, w, sf, seed = (int(v) for v in G10[name + "_dims"])
    fc, at = unet_state(6, 4, 10 + seed, head_gain=20.0), unet_state(20, 5, 20 + seed, head_gain=5.0)
    wsum = [sum(float(v.double().abs().sum()) for v in sd.values()) for sd in (fc, at)]
    assert np.allclose(wsum, G10[name + "_wsum"], rtol=1e-12), "the seeded weights differ from the ones the fixture was made with"
    f0, f1 = frame_pair(seed, h, w)
    assert np.array_equal(f0.numpy(), G10[name + "_frame0"]) and np.array_equal(f1.numpy(), G10[name + "_frame1"])
    return h, w, sf, fc, at, f0, f1


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_oracle_matches_reference_frames(name):
    h, w, sf, fc, at, f0, f1 = _case(name)
    out, outf = interp_ref.interpolate_pair(fc, at, f0, f1, sf, return_float=True)
    exp = G10[name + "_out"]
    assert out.shape == exp.shape == (sf - 1, h, w, 3)
    # same torch, same machine class, same operation order: the uint8 frames are identical
    assert np.array_equal(out.numpy(), exp)
    if name + "_float" in G10:
        mean = torch.tensor(interp_ref.MEAN).view(1, 3, 1, 1)
        assert np.allclose((outf + mean).numpy(), G10[name + "_float"], atol=1e-6)


def test_weight_container_is_the_reference_architecture():
    from avtex import slowmo
    h, w, sf, fc, at, f0, f1 = _case("b")
    net = slowmo.UNet(6, 4)
    assert set(net.state_dict().keys()) == set(fc.keys())
    net.load_state_dict(fc)
    x = torch.cat((interp_ref.to_tensor(f0), interp_ref.to_tensor(f1)), 0).unsqueeze(0)
    with torch.no_grad():
        flow = net(x)[0]
    assert np.allclose(flow.numpy(), G10["b_flow"], atol=1e-5)
    assert set(slowmo.UNet(20, 5).state_dict().keys()) == set(at.keys())


def test_interpolator_rejects_what_the_reference_cannot_do():
    from avtex import slowmo
    from avtex._lib import AvtError
    with pytest.raises(AvtError):
        slowmo.Interpolator(16, 16, 5, "cpu")   # rounds down to 0 x 0 (interpolate.py:64-66)
    with pytest.raises(AvtError):
        slowmo.Interpolator(64, 64, 1, "cpu")
