"""Whole-model parity for the interaction models (FFM, PNN, NFM, AFM, DCN, and the thin compositions DNN,
WideDeep): logits and two BaseModel.train_step updates on the MI355X against the CPU oracle carrying the same
initial weights (oracle/ctr_oracle.py; torch autograd + clip_grad_norm_ + torch optimizer)."""
import pytest
import torch

from oracle import ctr_oracle as O

pytestmark = pytest.mark.gpu

VOCAB = [11, 7, 50, 3, 200, 23]
DIM, BATCH = 8, 48


def key_to_oracle(k, kind=None):
    """state_dict key of the device model -> key of OracleCTRModel (None: table the forward never reads)."""
    if kind == "deepcrossing" and k.startswith("mlp."):
        return k
    for lr_name in ("lr_layer", "wide", "linear"):
        p = f"{lr_name}.scalar.embedded_to_1_dim_features."
        if k.startswith(p):
            return "emb1." + k[len(p):]
        if k == f"{lr_name}.bias" and not k.startswith("linear.bias_"):
            return "bias"
        if k.startswith(f"{lr_name}.scalar.embedded_features."):
            return None
    if k.startswith("embedding_layer.embedded_features."):
        return "emb." + k[len("embedding_layer.embedded_features."):]
    if k.startswith("embedding_layer.embedded_to_1_dim_features."):
        return None
    if k.startswith("mlp."):
        return "deep." + k[4:]
    if k.startswith("field_aware.tables."):
        return "ffm." + k[len("field_aware.tables."):]
    if k.startswith("pool."):
        return k[5:]
    if k.startswith("cross."):
        return k[6:]
    return k


def noise_driven(kind, kw, k):
    """Parameters whose gradient is mathematically zero in a residual stack that ends in LayerNorm: the 1-wide last
    block (model/DeepCrossing/DeepCrossing.py:24) normalises over ONE feature, so its output is `beta` whatever fc
    and gamma are.  torch's CPU LayerNorm leaves rounding noise there (1e-9 .. 1e-6, see
    tests/golden/train_deepcrossing_adam_ln.npz), the kernels here give exact zeros, and Adam turns either into steps
    of up to lr -- there is no value to compare."""
    if kind != "deepcrossing" or kw.get("normalization") != "ln":
        return False
    last = len(kw["hidden_dim_list"])
    return k in (f"mlp.{last}.fc.weight", f"mlp.{last}.fc.bias", f"mlp.{last}.norm_layer.weight")


def build(kind, optimizer, **kw):
    from recommendsystem_amd import data, models
    from recommendsystem_amd.base_model import random_all
    random_all(5)
    rec = data.criteo_x1_recorder(embedding_dim=DIM, batch_size=BATCH, vocab_sizes=VOCAB)
    cls = {"ffm": models.FFM, "pnn": models.PNN, "nfm": models.NFM, "afm": models.AFM, "dcn": models.DCN,
           "dnn": models.DNN, "widedeep": models.WideDeep, "deepcrossing": models.DeepCrossing}[kind]
    m = cls(rec, 1e-2, 0, optimizer_type=optimizer, **kw)
    with torch.no_grad():                       # rows large enough for every term to matter
        for n, p in m.named_parameters():
            if "embedded" in n or "tables" in n:
                p[1:].normal_(0, 0.3)
            if n == "v" or n.startswith("cross."):
                p.mul_(0.2)
    m.train()
    return m, rec


def make_oracle(kind, m, rec, kw):
    okw = {}
    if "hidden_dim_list" in kw:
        okw["hidden"] = kw["hidden_dim_list"]
    if kind not in ("ffm", "afm"):
        okw["normalization"] = kw.get("normalization", "bn")
    for a, b in (("pnn_type", "pnn_type"), ("hidden_dim", "nfm_hidden_dim"), ("attention_dim", "attention_dim"),
                 ("cross_layers_num", "cross_layers_num"), ("activation_func", "activation")):
        if a in kw:
            okw[b] = kw[a]
    o = O.OracleCTRModel(kind, rec.feature_map, rec.features, DIM, **okw)
    sd = {}
    for k, v in m.state_dict().items():
        ok = key_to_oracle(k, kind)
        if ok is not None:
            sd[ok] = v.detach().cpu().clone()
    # NFM's Linear over index values is called `linear` in both; its bias must not be taken for the LR bias
    if kind == "nfm":
        sd["linear.bias"] = m.state_dict()["linear.bias"].cpu().clone()
        sd.pop("bias", None)
    missing, unexpected = o.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k == "bias" or k.startswith("emb1.") for k in missing), missing
    o.train()
    return o


CASES = [
    ("ffm", "AdamW", {}),
    ("pnn", "Adam", dict(hidden_dim_list=[32, 16], pnn_type="inner", normalization=None)),
    ("pnn", "SGD", dict(hidden_dim_list=[32, 16], pnn_type="outer", normalization="bn")),
    ("pnn", "Adam", dict(hidden_dim_list=[32, 16], pnn_type="both", normalization="ln")),
    ("nfm", "SGD", dict(hidden_dim_list=[32, 16], hidden_dim=4, normalization="bn")),
    ("nfm", "Adam", dict(hidden_dim_list=[32, 16], hidden_dim=4, normalization=None)),
    ("afm", "Adam", dict(attention_dim=8)),
    ("afm", "SGD", dict(attention_dim=4)),
    # (BatchNorm makes the preceding bias gradient mathematically zero; Adam would amplify its rounding noise
    #  into full steps, so the bn cases use SGD)
    ("dcn", "SGD", dict(hidden_dim_list=[32, 16], cross_layers_num=3, normalization="bn")),
    ("dcn", "Adam", dict(hidden_dim_list=[32, 16], cross_layers_num=2, normalization=None)),
    ("dnn", "SGD", dict(hidden_dim_list=[32, 16], normalization="bn")),
    ("dnn", "Adam", dict(hidden_dim_list=[32, 16], normalization="ln")),
    ("widedeep", "AdamW", dict(hidden_dim_list=[32, 16], normalization="ln")),
    # model/DeepCrossing/DeepCrossing.py:15-36: residual blocks all the way down to the 1-wide one
    ("deepcrossing", "SGD", dict(hidden_dim_list=[32, 16], normalization="bn")),
    ("deepcrossing", "Adam", dict(hidden_dim_list=[32, 32], normalization="ln", activation_func="tanh")),
    ("deepcrossing", "AdamW", dict(hidden_dim_list=[48, 16], normalization="ln", activation_func="leakyrelu")),
]


@pytest.mark.parametrize("kind,optimizer,kw", CASES, ids=[f"{c[0]}-{c[1]}-{i}" for i, c in enumerate(CASES)])
def test_model_logits_and_train_steps(kind, optimizer, kw):
    m, rec = build(kind, optimizer, **kw)
    o = make_oracle(kind, m, rec, kw)
    opt = O.make_optimizer(optimizer, o.parameters(), 1e-2)
    g = torch.Generator().manual_seed(77)
    for step in range(2):
        x = {c: (torch.rand(BATCH, generator=g) ** 2 * v).long().clamp(0, v - 1) for c, v in zip(rec.features, VOCAB)}
        y = (torch.rand(BATCH, generator=g) < 0.4).long()
        with torch.no_grad():                            # both sides run this extra training-mode forward
            want = o(x)
            got = m.logits(x)
        torch.testing.assert_close(got.cpu(), want, rtol=2e-5, atol=2e-6, msg=lambda s: f"logits step {step}: {s}")
        loss = m.train_step(x, {"label": y})
        ref_loss, _ = O.train_step(o, opt, x, y)
        torch.testing.assert_close(loss.detach().cpu().view(()), ref_loss.view(()), rtol=2e-5, atol=1e-6)
        osd = o.state_dict()
        for k, v in m.state_dict().items():
            ok = key_to_oracle(k, kind)
            if kind == "nfm" and k == "linear.bias":
                ok = "linear.bias"
            if ok is None or not v.dtype.is_floating_point or noise_driven(kind, kw, k):
                continue
            # Adam divides by sqrt(v): where a gradient is itself at rounding level the step (lr = 1e-2) amplifies the
            # difference in summation order, so its absolute floor is a fraction of lr
            atol = 5e-5 if optimizer in ("Adam", "AdamW") else 5e-6
            torch.testing.assert_close(v.cpu(), osd[ok], rtol=1e-4, atol=atol, msg=lambda s, k=k: f"{k} after step {step}: {s}")
    m.embedding_layer._tab.layout.check_indices()
