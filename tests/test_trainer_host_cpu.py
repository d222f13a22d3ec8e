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


def test_bench_starts_its_own_ranks_when_run_without_a_launcher():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment: the script starts the two rank
    processes itself (before anything touches a GPU), they rendezvous on 127.0.0.1 and rank 0 prints one JSON line.
    --launch-check stops after the rendezvous and one gloo all-reduce, so this runs without a GPU."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--launch-check", "1"], env=env, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out == {"launch_check": "ok", "world": 2, "rank_mask": 3}


def test_bench_rank_failure_ends_the_launch():
    """A rank that exits with an error takes the launch down with its status instead of leaving the others waiting."""
    import os
    import subprocess
    import sys
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU (the ranks are meant to fail)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    # without --launch-check the ranks need a GPU, which this container does not have: every rank exits non-zero
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=180)
    assert r.returncode != 0 and "needs an MI355X" in r.stderr


def test_batch_statistics_of_the_union_from_per_rank_statistics():
    """What BatchNorm on several GPUs all-gathers and combines (ops.combine_batch_stats) against the statistics of the
    concatenated batch, with a large common offset (where E[x^2] - E[x]^2 would lose every digit)."""
    import torch
    from recommendsystem_amd import ops
    g = torch.Generator().manual_seed(0)
    parts = [torch.randn(64, 9, generator=g, dtype=torch.float64) * (0.1 + r) + 1000.0 * (r % 2) + 5e3 for r in range(4)]
    means = torch.stack([p.mean(0) for p in parts]).float()
    variances = torch.stack([p.var(0, unbiased=False) for p in parts]).float()
    mean, var = ops.combine_batch_stats(means, variances)
    union = torch.cat(parts, 0)
    torch.testing.assert_close(mean.double(), union.mean(0), rtol=1e-6, atol=0)
    torch.testing.assert_close(var.double(), union.var(0, unbiased=False), rtol=1e-5, atol=0)
