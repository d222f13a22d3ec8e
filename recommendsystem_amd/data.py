"""Batch plumbing around the hot path: DataRecorder-shaped configuration objects, synthetic
Criteo_x1-shaped batches, and a columnar feeder.

The reference's DataRecorder (model/utils/PreProcess.py:19-59) is host-side ETL and is out of
scope; the layers only read four attributes from it (``embedding_dim``, ``feature_map``,
``features``, ``input_dim``), which ``Recorder`` carries.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch


class Recorder:
    """The attributes of the reference's DataRecorder that the model code reads
    (PreProcess.py:25-27,46-48; Layers.py:115-117,143-145)."""

    def __init__(self, feature_map: Dict[str, dict], features: Sequence[str], embedding_dim: int, batch_size: int,
                 label_col: str = "label"):
        self.feature_map = feature_map
        self.features = list(features)
        self.embedding_dim = embedding_dim
        self.batch_size = batch_size
        self.categorical_col = [c for c in self.features if feature_map[c]["type"] == "categorical"]
        self.numerical_col = [c for c in self.features if feature_map[c]["type"] == "numerical"]
        self.feature_num = len(self.features)
        self.input_dim = self.feature_num * embedding_dim        # PreProcess.py:48
        self.label_col = label_col
        self.train_loader = self.valid_loader = self.test_loader = None


def criteo_x1_vocab_sizes(n_small: int = 13, n_large: int = 26, small_vocab: int = 100, lo: float = 10.0,
                          hi: float = 1.0e6) -> List[int]:
    """Criteo_x1-shaped vocabularies: 13 pre-bucketised "I" fields with small vocabularies and 26 "C"
    fields log-spaced from 10 to 1,000,000 rows (the real per-field sizes are not published with
    the reference; these are this repository's stand-ins and are printed with every result)."""
    large = [int(round(lo * (hi / lo) ** (i / (n_large - 1)))) for i in range(n_large)]
    return [small_vocab] * n_small + large


def criteo_x1_recorder(embedding_dim: int = 16, batch_size: int = 4096, vocab_sizes: Optional[Sequence[int]] = None):
    vs = list(vocab_sizes) if vocab_sizes is not None else criteo_x1_vocab_sizes()
    names = [f"I{i + 1}" for i in range(13)] + [f"C{i + 1}" for i in range(26)]
    names = names[:len(vs)] if len(vs) <= 39 else [f"F{i}" for i in range(len(vs))]
    fm = {c: {"type": "categorical", "vocab_size": int(v)} for c, v in zip(names, vs)}
    return Recorder(fm, names, embedding_dim, batch_size)


class ZipfSampler:
    """Indices in [1, V-1] with P(rank k) ~ k^-s; rank 1 is index 1, the hottest row, matching the
    frequency-sorted vocabularies of the reference's Tokenizer (model/utils/Tokenizer.py:75-79).
    Index 0 (padding) is never drawn."""

    def __init__(self, vocab_sizes: Sequence[int], s: float = 1.1, device="cpu", uniform: bool = False):
        self.vocab = list(vocab_sizes)
        self.device = torch.device(device)
        self.uniform = uniform
        self.cdfs = []
        if not uniform:
            for v in self.vocab:
                k = torch.arange(1, max(v, 2), dtype=torch.float64)
                w = k.pow(-s)
                self.cdfs.append((torch.cumsum(w, 0) / w.sum()).to(torch.float32).to(self.device))

    def sample(self, batch: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """[F, batch] int64."""
        cols = []
        for f, v in enumerate(self.vocab):
            if v <= 2:
                cols.append(torch.ones(batch, dtype=torch.int64, device=self.device))
                continue
            if self.uniform:
                cols.append(torch.randint(1, v, (batch,), generator=generator, device=self.device))
            else:
                u = torch.rand(batch, generator=generator, device=self.device)
                cols.append((torch.searchsorted(self.cdfs[f], u) + 1).clamp_(1, v - 1))
        return torch.stack(cols, 0)


class ColumnarFeeder:
    """Pre-staged batches for the fused engine: one [F, B] int64 index matrix and one [B] label vector
    per step, already resident on the device (one H2D copy per step when fed from the host instead of
    the reference's one copy per column, model/utils/BaseModel.py:204-208)."""

    def __init__(self, idx_batches: List[torch.Tensor], label_batches: List[torch.Tensor], columns: Sequence[str],
                 label_col: str = "label"):
        self.idx = idx_batches
        self.labels = label_batches
        self.columns = list(columns)
        self.label_col = label_col

    def __len__(self):
        return len(self.idx)

    def __iter__(self):
        """Yields (inputs, labels) in the reference's dict-of-columns format (views, no copies)."""
        for idx, y in zip(self.idx, self.labels):
            inputs = {c: idx[i] for i, c in enumerate(self.columns)}
            inputs[self.label_col] = y                      # the dataset's feature dict carries every column
            yield inputs, {self.label_col: y}


def synthetic_batches(vocab_sizes, batch: int, n_batches: int, device, seed: int = 2021, zipf_s: float = 1.1,
                      uniform: bool = False, pos_rate: float = 0.25, planted: bool = False):
    """n_batches of (idx [F,B] int64, labels [B] int64) on `device`.

    planted=True draws labels from a fixed random logistic model of the indices so that a trained
    model has signal to find (used by the AUC parity test); otherwise labels are Bernoulli(pos_rate)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sampler = ZipfSampler(vocab_sizes, s=zipf_s, device="cpu", uniform=uniform)
    w = None
    if planted:
        w = [torch.randn(v, generator=g) * 0.7 for v in vocab_sizes]
    idxs, labels = [], []
    for _ in range(n_batches):
        idx = sampler.sample(batch, generator=g)
        if planted:
            z = sum(w[f][idx[f]] for f in range(len(vocab_sizes))) / (len(vocab_sizes) ** 0.5) - 1.0
            y = (torch.rand(batch, generator=g) < torch.sigmoid(z)).long()
        else:
            y = (torch.rand(batch, generator=g) < pos_rate).long()
        idxs.append(idx.to(device))
        labels.append(y.to(device))
    return idxs, labels


# -------------------------------------------------------------------------------------------------
# On-disk formats of the reference's preprocessing output
# -------------------------------------------------------------------------------------------------
def build_encoding_map(values) -> (dict, int):
    """Frequency-sorted vocabulary of one categorical column (model/utils/Tokenizer.py:70-82).

    Tokens are ranked by (-count, str(token)); index 0 is PAD (token -1), ranks start at 1, and one more index
    -- the last -- is reserved for out-of-vocabulary values (token -2).  Returns (map, vocab_size) with
    vocab_size counting PAD and OOV, which is the `vocab_size` written to feature_map.json."""
    from collections import Counter
    word_counts = sorted(((str(k), int(v)) for k, v in Counter(values).items()), key=lambda x: (-x[1], x[0]))
    enc = {-1: 0}
    enc.update({int(tok): i for i, (tok, _) in enumerate(word_counts, 1)})
    n = len(enc)
    enc[-2] = n
    return enc, n + 1


def encode_column(values, encoding_map: dict) -> torch.Tensor:
    """Token -> index with unseen tokens sent to the OOV index (the map's entry for -2)."""
    oov = encoding_map[-2]
    return torch.tensor([encoding_map.get(int(v), oov) for v in values], dtype=torch.int64)


class EncodedTable:
    """One split of the reference's encoded data: checkpoint/encoded_column/<split>/encoded_table.parquet
    (model/utils/PreProcess.py:61-72), held as ONE columnar int64 matrix [F, N] in pinned host memory plus the
    label vector, so a training batch is a contiguous [F, B] slice and reaches the GPU in one copy (the reference
    moves 40 per-column tensors per step, model/utils/BaseModel.py:204-208)."""

    def __init__(self, columns: Sequence[str], matrix: torch.Tensor, labels: torch.Tensor, label_col: str = "label"):
        self.columns = list(columns)
        self.matrix = matrix          # [F, N] int64
        self.labels = labels          # [N]
        self.label_col = label_col

    @classmethod
    def from_parquet(cls, path: str, columns: Sequence[str], label_col: str = "label", pin: bool = True):
        import pyarrow.parquet as pq
        table = pq.read_table(path, columns=list(columns) + [label_col])
        n = table.num_rows
        mat = torch.empty((len(columns), n), dtype=torch.int64)
        for i, c in enumerate(columns):
            mat[i] = torch.from_numpy(table.column(c).to_numpy(zero_copy_only=False).astype("int64", copy=True))
        lab = torch.from_numpy(table.column(label_col).to_numpy(zero_copy_only=False).astype("int64", copy=True))
        if pin and torch.cuda.is_available():
            mat, lab = mat.pin_memory(), lab.pin_memory()
        return cls(columns, mat, lab, label_col)

    def __len__(self):
        return self.matrix.shape[1]

    def loader(self, batch_size: int, device=None, drop_last: bool = False):
        return EncodedLoader(self, batch_size, device, drop_last)


class EncodedLoader:
    """Sequential batches of an EncodedTable in the reference's (inputs, labels) format.  With `device` set, every
    batch is moved with one non-blocking copy of its [F, B] slice and the per-column tensors are views of it."""

    def __init__(self, table: EncodedTable, batch_size: int, device=None, drop_last: bool = False):
        self.table, self.batch_size, self.device, self.drop_last = table, int(batch_size), device, drop_last

    def __len__(self):
        n = len(self.table)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        t = self.table
        for b in range(len(self)):
            lo, hi = b * self.batch_size, min((b + 1) * self.batch_size, len(t))
            idx, y = t.matrix[:, lo:hi], t.labels[lo:hi]
            if self.device is not None:
                idx = idx.contiguous().to(self.device, non_blocking=True)
                y = y.to(self.device, non_blocking=True)
            inputs = {c: idx[i] for i, c in enumerate(t.columns)}
            inputs[t.label_col] = y
            yield inputs, {t.label_col: y}


def load_encoded_checkpoint(root: str, embedding_dim: int, batch_size: int, device=None, label_col: str = "label",
                            splits=("train", "valid", "test")) -> Recorder:
    """Recorder over a reference checkpoint directory: `<root>/feature_map.json` (column -> {"type", "vocab_size"},
    Tokenizer.py:58-60) and `<root>/encoded_column/<split>/encoded_table.parquet`.  `features` is categorical
    columns then numerical ones (PreProcess.py:46); only categorical columns are read into the index matrix."""
    import json
    import os
    with open(os.path.join(root, "feature_map.json"), "r", encoding="utf-8") as f:
        feature_map = json.load(f)
    cat = [c for c, info in feature_map.items() if info["type"] == "categorical"]
    num = [c for c, info in feature_map.items() if info["type"] == "numerical"]
    rec = Recorder(feature_map, cat + num, embedding_dim, batch_size, label_col=label_col)
    for split in splits:
        path = os.path.join(root, "encoded_column", split, "encoded_table.parquet")
        if not os.path.exists(path):
            continue
        loader = EncodedTable.from_parquet(path, cat, label_col).loader(batch_size, device)
        setattr(rec, f"{split}_loader", loader)
    return rec
