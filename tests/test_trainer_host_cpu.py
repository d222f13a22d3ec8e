"""Host-side logic of the generic touched-rows trainer that needs no GPU: grouping of captured lookups into shared
sort plans, and the per-forward cache of index matrices."""
from types import SimpleNamespace

import pytest
import torch

from recommendsystem_amd import functional as Fn
from recommendsystem_amd._lib import RSError
from recommendsystem_amd.sparse_trainer import SparseTrainer


def rec(kind, holder, idx, with_grad=True):
    out = torch.zeros(3, requires_grad=True)
    if with_grad:
        out.grad = torch.ones(3)
    return {"kind": kind, "holder": holder, "idx": idx, "out": out}


def holder(vocab):
    return SimpleNamespace(layout=SimpleNamespace(vocab_sizes=list(vocab)))


def test_lookups_of_one_batch_share_a_plan():
    idx = torch.zeros(2, 3, dtype=torch.int64)
    h_emb, h_lr = holder([5, 7]), holder([5, 7])
    groups = SparseTrainer._groups([rec("lr", h_lr, idx), rec("emb", h_emb, idx)])
    assert len(groups) == 1 and set(groups[0]) == {"lr", "emb"}


def test_other_batch_or_other_vocabulary_gets_its_own_plan():
    idx, idx2 = torch.zeros(2, 3, dtype=torch.int64), torch.zeros(2, 3, dtype=torch.int64)
    groups = SparseTrainer._groups([rec("lr", holder([5, 7]), idx), rec("emb", holder([5, 7]), idx2)])
    assert len(groups) == 2
    groups = SparseTrainer._groups([rec("lr", holder([5, 7]), idx), rec("emb", holder([5, 8]), idx)])
    assert len(groups) == 2


def test_ffm_never_shares_with_embedding_rows_but_does_with_lr():
    idx = torch.zeros(2, 3, dtype=torch.int64)
    groups = SparseTrainer._groups([rec("lr", holder([5, 7]), idx), rec("ffm", holder([5, 7]), idx),
                                    rec("emb", holder([5, 7]), idx)])
    assert [set(g) for g in groups] == [{"lr", "ffm"}, {"emb"}]


def test_unused_lookup_is_skipped_and_double_lookup_is_an_error():
    idx = torch.zeros(2, 3, dtype=torch.int64)
    h = holder([5, 7])
    assert SparseTrainer._groups([rec("emb", h, idx, with_grad=False)]) == []
    with pytest.raises(RSError):
        SparseTrainer._groups([rec("emb", h, idx), rec("emb", h, idx)])


def test_index_matrix_is_built_once_per_forward():
    calls = []
    x = {"a": torch.tensor([1, 2]), "b": torch.tensor([3, 4])}

    def build():
        calls.append(1)
        return torch.stack([x["a"], x["b"]], 0)

    with Fn.TableCapture() as cap:
        m1 = cap.index_matrix(x, ["a", "b"], build)
        m2 = cap.index_matrix(x, ["a", "b"], build)
        m3 = cap.index_matrix(x, ["a"], build)
        m4 = cap.index_matrix(dict(x), ["a", "b"], build)          # another batch object
    assert m1 is m2 and m3 is not m1 and m4 is not m1 and len(calls) == 3
    assert Fn.TableCapture.active is None
    with pytest.raises(RuntimeError):
        with Fn.TableCapture():
            with Fn.TableCapture():
                pass
    assert Fn.TableCapture.active is None
