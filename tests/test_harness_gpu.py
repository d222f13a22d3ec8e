"""The train/eval loop end to end on the MI355X (fit -> train_epoch -> eval_step -> early stop -> reload best), on the
module path and on the fused engine, and the fused step at the benchmark's batch sizes against the CPU oracle."""
import os

import pytest
import torch

from oracle import ctr_oracle as O

pytestmark = pytest.mark.gpu

VOCAB = [50, 200, 1000, 30, 10, 500]


def loaders(n_train=24, n_valid=6, batch=128):
    from recommendsystem_amd import data
    idxs, labs = data.synthetic_batches(VOCAB, batch, n_train + n_valid, "cpu", seed=3, planted=True)
    rec = data.criteo_x1_recorder(embedding_dim=8, batch_size=batch, vocab_sizes=VOCAB)
    rec.train_loader = data.ColumnarFeeder(idxs[:n_train], labs[:n_train], rec.features)
    rec.valid_loader = data.ColumnarFeeder(idxs[n_train:], labs[n_train:], rec.features)
    return rec


@pytest.mark.parametrize("fused", [False, True])
def test_fit_runs_the_reference_loop(fused, tmp_path):
    from recommendsystem_amd import models
    from recommendsystem_amd.base_model import random_all
    random_all(2021)
    rec = loaders()
    m = models.DeepFM(rec, 1e-2, 0, hidden_dim_list=[32, 16], normalization=None, num_epochs=6, checkpoint_root=str(tmp_path))
    if fused:
        m.use_fused_step(mode="sparse", use_graph=True)
    m.fit()
    # the reference's comparison (lower AUC = better) keeps epoch 1 as best, decays the rate after every later epoch
    # and stops after `patience` = 2 of them: 3 epochs in all
    assert m.stop_training and m.epoch_index == 2
    path = os.path.join(m.checkpoint, "DeepFM.model")
    assert os.path.exists(path)
    best = torch.load(path, map_location="cpu", weights_only=True)
    for k, v in m.state_dict().items():             # fit() ends by reloading the saved best weights
        assert torch.equal(v.cpu(), best[k]), k
    auc = m.evaluate(rec.valid_loader)
    assert abs(auc - m.best_auc) < 1e-9 and 0.5 < auc < 1.0
    lr_now = m._engine.lr if fused else m.optimizer.param_groups[0]["lr"]
    assert abs(lr_now - 1e-4) < 1e-9                 # two decays of x0.1


def test_evaluate_matches_oracle_on_same_weights(tmp_path):
    from recommendsystem_amd import models
    from recommendsystem_amd.base_model import random_all
    random_all(7)
    rec = loaders(n_train=2, n_valid=8)
    m = models.FM(rec, 1e-2, 0, checkpoint_root=str(tmp_path))
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "embedded" in n:
                p[1:].normal_(0, 0.3)
    o = O.OracleCTRModel("fm", rec.feature_map, rec.features, 8)
    sd = {}
    for k, v in m.state_dict().items():
        if k.startswith("embedding_layer.embedded_features."):
            sd["emb." + k[len("embedding_layer.embedded_features."):]] = v.cpu()
        elif k.startswith("lr_layer.scalar.embedded_to_1_dim_features."):
            sd["emb1." + k[len("lr_layer.scalar.embedded_to_1_dim_features."):]] = v.cpu()
        elif k == "lr_layer.bias":
            sd["bias"] = v.cpu()
    o.load_state_dict(sd)
    o.eval()
    auc = m.evaluate(rec.valid_loader)
    with torch.no_grad():
        p = torch.cat([torch.sigmoid(o({c: x[c] for c in rec.features})).view(-1) for x, _ in rec.valid_loader])
    y = torch.cat([l["label"] for _, l in rec.valid_loader])
    assert abs(auc - O.auc(y, p)) < 1e-6 and abs(m.last_logloss - O.logloss(y, p)) < 1e-6


@pytest.mark.parametrize("batch", [4096, 16384])
def test_fused_step_at_benchmark_batch_sizes(batch):
    """B = 4096 (configs 2-4) and B = 16384 (config 5) on Criteo_x1-shaped fields (39 fields, vocabularies up to 1e5
    here to keep the CPU oracle quick): loss, logits and every parameter after two fused SGD steps (graph replay)."""
    from recommendsystem_amd import data, models
    from recommendsystem_amd.base_model import random_all
    random_all(11)
    vocab = [100] * 13 + [int(round(10 * (1e5 / 10) ** (i / 25))) for i in range(26)]
    rec = data.criteo_x1_recorder(embedding_dim=16, batch_size=batch, vocab_sizes=vocab)
    m = models.DeepFM(rec, 0.05, 0, hidden_dim_list=[64, 32], normalization=None, optimizer_type="SGD")
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "embedded" in n:
                p[1:].normal_(0, 0.05)
    m.train()
    o = O.OracleCTRModel("deepfm", rec.feature_map, rec.features, 16, hidden=[64, 32])
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
    o.train()
    opt = O.make_optimizer("SGD", o.parameters(), 0.05)
    eng = m.use_fused_step(mode="sparse", use_graph=True, max_norm=10.0)
    idxs, labs = data.synthetic_batches(vocab, batch, 2, "cpu", seed=5)
    for idx, y in zip(idxs, labs):
        x = {c: idx[i] for i, c in enumerate(rec.features)}
        with torch.no_grad():
            want = o(x)
        got = eng.logits(idx.cuda())
        torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-6)
        loss = eng.step_tensors(idx.cuda(), y.cuda())
        ref, _ = O.train_step(o, opt, x, y)
        torch.testing.assert_close(loss.cpu()[0], ref, rtol=1e-5, atol=1e-7)
    osd = o.state_dict()
    for k, v in m.state_dict().items():
        ok = ("emb." + k[len("embedding_layer.embedded_features."):] if k.startswith("embedding_layer.embedded_features.")
              else "emb1." + k[len("lr_layer.scalar.embedded_to_1_dim_features."):]
              if k.startswith("lr_layer.scalar.embedded_to_1_dim_features.") else "bias" if k == "lr_layer.bias"
              else k if k.startswith("deep.") else None)
        if ok is not None:
            torch.testing.assert_close(v.cpu(), osd[ok], rtol=2e-5, atol=1e-7, msg=lambda s, k=k: f"{k}: {s}")
    m.embedding_layer._tab.layout.check_indices()
