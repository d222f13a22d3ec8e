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


def _oracle_stack(blocks, x, g):
    """The same run of blocks on the CPU oracle (oracle/ctr_oracle.py OracleMLPBlock = Layers.py:8-47, dropout 0)."""
    from oracle import ctr_oracle as O
    act_name = {1: "relu", 2: "tanh", 3: "leakyrelu", 4: "prelu"}
    obs = []
    for b in blocks:
        ob = O.OracleMLPBlock(b.fc.in_features, b.fc.out_features, act_name[b._act], b.normalization or None)
        ob.load_state_dict({k.replace("activation_func.", "prelu."): v.detach().cpu() for k, v in b.state_dict().items()})
        obs.append(ob.train())
    xo = x.detach().cpu().requires_grad_(True)
    h = xo
    for ob in obs:
        h = ob(h)
    (h * g.cpu()).sum().backward()
    return h.detach(), xo.grad, obs


@pytest.mark.parametrize("spec", [
    [(40, 24, "relu", None), (24, 12, "tanh", None), (12, 1, "relu", None)],            # all fused, 1-wide head
    [(40, 24, "leakyrelu", None), (24, 12, "relu", "bn"), (12, 8, "tanh", None)],         # a normalised block in the middle
    [(40, 24, "prelu", None), (24, 12, "relu", "ln"), (12, 1, "tanh", None)],             # PReLU cannot ride in an epilogue
    [(1365, 64, "relu", None), (64, 1, "relu", None)],                                     # odd input width: padded to 1368
], ids=["plain-head", "bn-middle", "prelu-ln-head", "odd-width"])
def test_mlp_stack_matches_oracle_and_block_by_block(spec):
    """layers.MLPStack (one autograd node, fused backward) against the CPU oracle's blocks and against the same
    modules run one by one."""
    from torch import nn
    from recommendsystem_amd import layers
    torch.manual_seed(3)
    blocks = [layers.MLPBlock(i, o, a, n) for i, o, a, n in spec]
    stack = layers.MLPStack(*blocks).cuda().train()
    B = 512
    x = torch.randn(B, spec[0][0], device="cuda").requires_grad_(True)
    g = torch.randn(B, spec[-1][1], device="cuda")
    want_y, want_dx, obs = _oracle_stack(blocks, x, g)
    y = stack(x)
    scale = max(1.0, float(want_y.abs().max()))
    torch.testing.assert_close(y.detach().cpu(), want_y, rtol=2e-5, atol=5e-6 * scale)
    (y * g).sum().backward()
    torch.testing.assert_close(x.grad.cpu(), want_dx, rtol=2e-5, atol=5e-6 * max(1.0, float(want_dx.abs().max())))
    for b, ob in zip(blocks, obs):
        oq = dict(ob.named_parameters())
        for k, p in b.named_parameters():
            q = oq[k.replace("activation_func.", "prelu.")]
            atol = 1e-5 * max(1.0, float(q.grad.abs().max()))
            if k == "fc.bias" and b.normalization:
                # a bias in front of a normalisation has gradient zero: both sides hold the rounding of a 512-term sum
                atol = 1e-4
            torch.testing.assert_close(p.grad.cpu(), q.grad, rtol=2e-5, atol=atol, msg=lambda s, k=k: f"d{k}: {s}")
    # block by block (nn.Sequential's own forward): same forward bits
    fused = {k: p.grad.clone() for k, p in stack.named_parameters()}
    dx_fused = x.grad.clone()
    for p in stack.parameters():
        p.grad = None
    x.grad = None
    # (BatchNorm running statistics were advanced once: irrelevant to the train-mode output)
    y2 = nn.Sequential.forward(stack, x)
    assert torch.equal(y2, y)
    (y2 * g).sum().backward()
    torch.testing.assert_close(x.grad, dx_fused, rtol=1e-5, atol=1e-6 * max(1.0, float(dx_fused.abs().max())))
    for k, p in stack.named_parameters():
        atol = 1e-4 if k.endswith("fc.bias") and blocks[int(k.split(".")[0])].normalization else \
            2e-6 * max(1.0, float(fused[k].abs().max()))
        torch.testing.assert_close(p.grad, fused[k], rtol=1e-5, atol=atol, msg=lambda s, k=k: f"d{k}: {s}")


def test_bce_of_logits_matches_torch():
    """functional.bce_of_logits = torch.sigmoid + F.binary_cross_entropy (mean), value and gradient, including
    saturated logits (torch clamps the logarithms at -100)."""
    from recommendsystem_amd import functional as Fn
    torch.manual_seed(0)
    z = (torch.randn(4096) * 4).tolist()
    z[:4] = [60.0, -60.0, 120.0, -120.0]
    z = torch.tensor(z)
    y = (torch.rand(4096) < 0.3).float()
    zc = z.clone().requires_grad_(True)
    want = torch.nn.functional.binary_cross_entropy(torch.sigmoid(zc), y)
    want.backward()
    for assume in (None, 1.0):
        zg = z.cuda().requires_grad_(True)
        got = Fn.bce_of_logits(zg.view(-1, 1), y.cuda(), assume_grad=assume)
        got.backward()
        torch.testing.assert_close(got.cpu(), want.detach(), rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(zg.grad.cpu(), zc.grad, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("B,K1,K2", [(4096, 624, 256), (37, 5, 3), (300, 64, 7), (1000, 700, 324), (16, 4, 4)])
def test_one_wide_linear_over_two_blocks(B, K1, K2):
    """functional.linear1_of_parts -- DCN's Linear(D + h_last, 1) over cat([cross, deep], 1) (DCN.py:31,52) without the
    concatenation -- against torch on the concatenated input: output and every gradient."""
    from recommendsystem_amd import functional as Fn
    g = torch.Generator().manual_seed(B + K1)
    x1, x2 = torch.randn(B, K1, generator=g), torch.randn(B, K2, generator=g)
    W, b = torch.randn(1, K1 + K2, generator=g) * 0.1, torch.randn(1, generator=g)
    gout = torch.randn(B, 1, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (x1, x2, W, b)]
    ref = torch.nn.functional.linear(torch.cat(ref_in[:2], 1), ref_in[2], ref_in[3])
    ref.backward(gout)
    ins = [t.to("cuda:0").requires_grad_(True) for t in (x1, x2, W, b)]
    out = Fn.linear1_of_parts(*ins)
    assert out is not None and out.shape == (B, 1)
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=2e-5, atol=2e-5)
    out.backward(gout.to("cuda:0"))
    for name, a, r in zip(("dx1", "dx2", "dW", "db"), ins, ref_in):
        torch.testing.assert_close(a.grad.cpu(), r.grad, rtol=1e-4, atol=1e-5 * max(1.0, float(r.grad.abs().max())), msg=name)
