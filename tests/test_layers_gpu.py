"""MLPBlock / ResidualMLPBlock on the MI355X against the vectors the reference's own blocks produced
(tests/golden/mlp_blocks.npz, written by oracle/gen_golden.py from model/utils/Layers.py:8-94): output, input
gradient, every parameter gradient, BatchNorm running statistics and the eval-mode output.  Tolerance: 1e-5
relative on fp32 with an absolute floor at the magnitude of the summed terms (inputs are N(0,1), 12 terms per dot)."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

NAMES = json.loads(str(load_golden("mlp_blocks.npz")["names"]))
RTOL, ATOL = 1e-5, 2e-6


def T(a):
    return torch.from_numpy(np.array(a))


def build(name, g):
    from recommendsystem_amd import layers
    parts = name.split("_")
    act, norm = parts[1], (None if parts[2] == "None" else parts[2])
    p = {k[len(name) + 3:]: T(v) for k, v in g.items() if k.startswith(name + ".p.")}
    if parts[0] == "mlp":
        blk = layers.MLPBlock(p["fc.weight"].shape[1], p["fc.weight"].shape[0], act, norm)
    else:
        blk = layers.ResidualMLPBlock(int(parts[3]), int(parts[4]), act, norm)
    own = {k for k, _ in blk.named_parameters()}
    assert own == set(p), own ^ set(p)                      # the reference's parameter names
    blk.load_state_dict(p, strict=False)
    return blk.cuda().train(), p


@pytest.mark.parametrize("name", NAMES)
def test_block_matches_reference_vectors(name):
    g = load_golden("mlp_blocks.npz")
    blk, p = build(name, g)
    x = T(g[name + ".x"]).cuda().requires_grad_(True)
    y = blk(x)
    torch.testing.assert_close(y.detach().cpu(), T(g[name + ".y"]), rtol=RTOL, atol=ATOL)
    (y * T(g[name + ".g"]).cuda()).sum().backward()
    torch.testing.assert_close(x.grad.cpu(), T(g[name + ".dx"]), rtol=RTOL, atol=ATOL, msg=lambda s: f"dx: {s}")
    for k, v in blk.named_parameters():
        torch.testing.assert_close(v.grad.cpu(), T(g[f"{name}.dp.{k}"]), rtol=RTOL, atol=ATOL,
                                   msg=lambda s, k=k: f"d{k}: {s}")
    if name + ".running_mean" in g:
        torch.testing.assert_close(blk.norm_layer.running_mean.cpu(), T(g[name + ".running_mean"]), rtol=RTOL, atol=1e-7)
        torch.testing.assert_close(blk.norm_layer.running_var.cpu(), T(g[name + ".running_var"]), rtol=RTOL, atol=1e-7)
        assert int(blk.norm_layer.num_batches_tracked) == 1
        blk.eval()
        with torch.no_grad():
            torch.testing.assert_close(blk(x).cpu(), T(g[name + ".y_eval"]), rtol=RTOL, atol=ATOL)


def test_residual_prelu_is_rejected_like_the_reference_fails():
    """The reference stores the nn.PReLU class (Layers.py:60) and fails at forward; here the constructor says so."""
    from recommendsystem_amd import layers
    with pytest.raises(ValueError):
        layers.ResidualMLPBlock(6, 6, "prelu", "bn")
    with pytest.raises(ValueError):
        layers.ResidualMLPBlock(6, 6, "relu", None)         # Layers.py:68-73: normalisation is mandatory


@pytest.mark.parametrize("B,i,o,act,norm", [(64, 624, 256, "relu", "bn"), (130, 256, 256, "leakyrelu", "ln"),
                                            (33, 40, 1, "tanh", "bn")])
def test_residual_block_against_oracle_at_model_widths(B, i, o, act, norm):
    """Widths of the DeepCrossing stack (624 -> ... -> 1), against the CPU oracle's block."""
    from oracle import ctr_oracle as O
    from recommendsystem_amd import layers
    torch.manual_seed(B + i)
    ref = O.OracleResidualBlock(i, o, act, norm).train()
    with torch.no_grad():
        ref.norm_layer.weight.uniform_(0.5, 1.5)
        ref.norm_layer.bias.normal_(0, 0.1)
    blk = layers.ResidualMLPBlock(i, o, act, norm)
    blk.load_state_dict({k: v for k, v in ref.state_dict().items()}, strict=True)
    blk = blk.cuda().train()
    x = torch.randn(B, i)
    gout = torch.randn(B, o)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(gout)
    xg = x.cuda().requires_grad_(True)
    y = blk(xg)
    y.backward(gout.cuda())
    torch.testing.assert_close(y.detach().cpu(), yr.detach(), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(xg.grad.cpu(), xr.grad, rtol=2e-5, atol=2e-5)
    rp = dict(ref.named_parameters())
    for k, v in blk.named_parameters():
        # floor: rounding of sums of B terms of order |gout| (BatchNorm makes the gradient of fc.bias mathematically
        # zero: what is left on either side is exactly that rounding)
        scale = float(rp[k].grad.abs().max()) + B * 1e-2
        torch.testing.assert_close(v.grad.cpu(), rp[k].grad, rtol=1e-4, atol=2e-5 * scale, msg=lambda s, k=k: f"d{k}: {s}")
    torch.testing.assert_close(blk.norm_layer.running_mean.cpu(), ref.norm_layer.running_mean, rtol=1e-5, atol=1e-6) \
        if norm == "bn" else None
