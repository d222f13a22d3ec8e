"""On-disk formats of the reference's preprocessing output (feature_map.json, encoded_table.parquet) and its
vocabulary rules, read without the reference's polars dependency."""
import json
import os

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import torch


def test_encoding_map_follows_the_reference_rules():
    from recommendsystem_amd import data
    values = [7, 7, 7, 3, 3, 10, 2, 2, 2, 2, 11]
    enc, vocab_size = data.build_encoding_map(values)
    # frequency first, then the STRING order of the token ("10" < "11" < "3"); PAD = 0, OOV = last index
    assert enc == {-1: 0, 2: 1, 7: 2, 3: 3, 10: 4, 11: 5, -2: 6} and vocab_size == 7
    got = data.encode_column([2, 11, 999, 7], enc)
    assert got.tolist() == [1, 5, 6, 2] and got.dtype == torch.int64
    assert int(got.max()) < vocab_size                      # every index addresses a row of nn.Embedding(vocab_size, d)


def write_checkpoint(root, n_train=50, n_valid=20):
    g = np.random.default_rng(0)
    fmap = {"userid": {"type": "categorical", "vocab_size": 26}, "pid": {"type": "categorical", "vocab_size": 4},
            "price": {"type": "numerical"}, "cate_id": {"type": "categorical", "vocab_size": 49}}
    os.makedirs(root, exist_ok=True)
    json.dump(fmap, open(os.path.join(root, "feature_map.json"), "w"))
    tables = {}
    for split, n in (("train", n_train), ("valid", n_valid)):
        cols = {"userid": g.integers(0, 26, n), "pid": g.integers(0, 4, n), "price": g.random(n),
                "cate_id": g.integers(0, 49, n), "label": g.integers(0, 2, n)}
        d = os.path.join(root, "encoded_column", split)
        os.makedirs(d, exist_ok=True)
        pq.write_table(pa.table(cols), os.path.join(d, "encoded_table.parquet"))
        tables[split] = cols
    return fmap, tables


def test_checkpoint_directory_to_recorder_and_batches(tmp_path):
    from recommendsystem_amd import data
    fmap, tables = write_checkpoint(str(tmp_path))
    rec = data.load_encoded_checkpoint(str(tmp_path), embedding_dim=4, batch_size=16)
    assert rec.features == ["userid", "pid", "cate_id", "price"]           # categorical first, then numerical
    assert rec.categorical_col == ["userid", "pid", "cate_id"] and rec.numerical_col == ["price"]
    assert rec.input_dim == 4 * 4 and rec.feature_map == fmap               # PreProcess.py:48 counts every feature
    assert rec.test_loader is None and len(rec.train_loader) == 4 and len(rec.valid_loader) == 2
    seen = 0
    for inputs, labels in rec.train_loader:
        b = labels["label"].numel()
        assert set(inputs) == {"userid", "pid", "cate_id", "label"}
        for c in ("userid", "pid", "cate_id"):
            assert inputs[c].dtype == torch.int64
            assert inputs[c].tolist() == tables["train"][c][seen:seen + b].tolist()
        assert labels["label"].tolist() == tables["train"]["label"][seen:seen + b].tolist()
        # one columnar matrix behind the per-column tensors: consecutive columns are b elements apart
        assert inputs["pid"].data_ptr() - inputs["userid"].data_ptr() == 8 * len(rec.train_loader.table)
        seen += b
    assert seen == 50                                                       # ragged last batch (50 = 3*16 + 2)
    assert sum(l["label"].numel() for _, l in rec.valid_loader) == 20


def test_layers_accept_the_loaded_recorder(tmp_path):
    """The modules read only embedding_dim / feature_map / features / input_dim (Layers.py:115-117,143-145)."""
    from recommendsystem_amd import data, layers
    write_checkpoint(str(tmp_path))
    rec = data.load_encoded_checkpoint(str(tmp_path), embedding_dim=4, batch_size=16)
    emb = layers.Embedding_layer(rec)
    assert emb.lookup_columns == ["userid", "pid", "cate_id"]               # numerical columns are skipped
    assert emb.embedded_features["cate_id"].weight.shape == (49, 4)
    assert float(emb.embedded_features["pid"].weight.detach()[0].abs().sum()) == 0.0  # PAD row
