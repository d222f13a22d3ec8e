"""The reference's on-disk formats driving the HIP path: a checkpoint directory as its preprocessing writes it
(feature_map.json + encoded_column/<split>/encoded_table.parquet, model/utils/PreProcess.py:61-72,
model/utils/Tokenizer.py:58-60) is read by data.load_encoded_checkpoint and trained on through BaseModel.fit() -- module
path and fused engine -- with the first steps checked against the CPU oracle on the same rows."""
import pytest
import torch

from oracle import ctr_oracle as O
from test_data_cpu import write_checkpoint

pytestmark = pytest.mark.gpu


def _oracle_for(m, rec, kind, hidden):
    o = O.OracleCTRModel(kind, rec.feature_map, rec.features, rec.embedding_dim, hidden=hidden, normalization=None)
    sd = {}
    for k, v in m.state_dict().items():
        if k.startswith("embedding_layer.embedded_features."):
            sd["emb." + k[len("embedding_layer.embedded_features."):]] = v.cpu().clone()
        elif k.startswith("lr_layer.scalar.embedded_to_1_dim_features."):
            sd["emb1." + k[len("lr_layer.scalar.embedded_to_1_dim_features."):]] = v.cpu().clone()
        elif k == "lr_layer.bias":
            sd["bias"] = v.cpu().clone()
        elif k.startswith("deep."):
            sd[k] = v.cpu().clone()
    o.load_state_dict(sd)
    return o.train()


@pytest.mark.parametrize("fused", [False, True])
def test_encoded_checkpoint_feeds_the_training_step(tmp_path, fused):
    from recommendsystem_amd import data, models
    from recommendsystem_amd.base_model import random_all
    fmap, tables = write_checkpoint(str(tmp_path), n_train=200, n_valid=64)
    rec = data.load_encoded_checkpoint(str(tmp_path), embedding_dim=8, batch_size=32, device="cuda:0")
    assert rec.categorical_col == ["userid", "pid", "cate_id"]            # the numerical column is not embedded (Layers.py:145)
    random_all(3)
    m = models.DeepFM(rec, 0.05, 0, hidden_dim_list=[16, 8], normalization=None, optimizer_type="SGD", num_epochs=2,
                      checkpoint_root=str(tmp_path / "ckpt"))
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "embedded" in n:
                p[1:].normal_(0, 0.2)
    o = _oracle_for(m, rec, "deepfm", [16, 8])
    opt = O.make_optimizer("SGD", o.parameters(), 0.05)
    if fused:
        m.use_fused_step(mode="sparse", use_graph=True)
    m.train()
    # the first three batches, step by step, against the oracle fed from the same parquet rows
    it = iter(rec.train_loader)
    for step in range(3):
        inputs, labels = next(it)
        lo, hi = step * 32, step * 32 + 32
        x = {c: torch.from_numpy(tables["train"][c][lo:hi].astype("int64")) for c in rec.categorical_col}
        y = torch.from_numpy(tables["train"]["label"][lo:hi].astype("int64"))
        assert inputs["userid"].is_cuda and inputs["userid"].cpu().tolist() == x["userid"].tolist()
        loss = m.train_step(inputs, labels)
        ref, _ = O.train_step(o, opt, x, y)
        torch.testing.assert_close(loss.detach().cpu().view(-1)[0], ref, rtol=2e-5, atol=1e-6)
    for c in rec.categorical_col:
        torch.testing.assert_close(m.state_dict()[f"embedding_layer.embedded_features.{c}.weight"].cpu(),
                                   o.emb[c].weight.detach(), rtol=2e-5, atol=1e-6)
    # and the whole loop: epochs over the ragged loader (200 = 6 * 32 + 8), evaluation on the valid split, checkpoint
    m.fit()
    auc = m.evaluate(rec.valid_loader)
    assert 0.0 <= auc <= 1.0 and m.last_logloss > 0
