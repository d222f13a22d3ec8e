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
