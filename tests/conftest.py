import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def golden_feature_map(g):
    return json.loads(str(g["feature_map"])), json.loads(str(g["features"]))


def golden_batch(g, features, prefix="x."):
    return {c: torch.from_numpy(g[prefix + c]) for c in features}


def oracle_model_from_golden(g, mirror_unused=True):
    """Build oracle.ctr_oracle.OracleCTRModel with the initial weights of a train_*.npz fixture."""
    from oracle import ctr_oracle as O
    fm, feats = golden_feature_map(g)
    kind = str(g["kind"])
    hidden = json.loads(str(g["hidden"]))
    norm = json.loads(str(g["normalization"]))
    act = str(g["activation"]) if "activation" in g else "ReLU"
    m = O.OracleCTRModel(kind, fm, feats, int(g["dim"]), hidden=hidden, normalization=norm, activation=act,
                         mirror_unused=mirror_unused)
    load_oracle_state(m, g, "init.")
    return m


# golden (reference state_dict) key -> oracle key
def _map_key(k):
    if k.startswith("embedding_layer.embedded_features."):
        return "emb." + k[len("embedding_layer.embedded_features."):]
    if k.startswith("embedding_layer.embedded_to_1_dim_features."):
        return "unused_emb1." + k[len("embedding_layer.embedded_to_1_dim_features."):]
    if k.startswith("lr_layer.scalar.embedded_features."):
        return "unused_emb." + k[len("lr_layer.scalar.embedded_features."):]
    if k.startswith("lr_layer.scalar.embedded_to_1_dim_features."):
        return "emb1." + k[len("lr_layer.scalar.embedded_to_1_dim_features."):]
    if k == "lr_layer.bias":
        return "bias"
    return k


def load_oracle_state(m, g, prefix):
    sd = {}
    for k, v in g.items():
        if k.startswith(prefix):
            sd[_map_key(k[len(prefix):])] = torch.from_numpy(np.array(v))
    own = m.state_dict()
    missing = [k for k in own if k not in sd]
    assert not missing, missing
    m.load_state_dict({k: sd[k] for k in own})


def golden_state(g, prefix):
    return {_map_key(k[len(prefix):]): torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith(prefix)}
