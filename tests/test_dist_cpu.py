"""Multi-process (gloo) rehearsal of the row-sharded embedding exchange on the CPU: the host protocol of
recommendsystem_amd.dist (ShardSpec, Comm, ShardedExchange) with torch-CPU stand-ins for the HIP kernels."""
import os

import pytest
import torch

from dist_launch import run_ranks


def test_shard_spec_partitions_every_row_once():
    from recommendsystem_amd.dist import ShardSpec
    vocab = [1, 2, 7, 8, 9, 1000, 3]
    for N in (1, 2, 3, 8):
        specs = [ShardSpec(vocab, N, r) for r in range(N)]
        for f, v in enumerate(vocab):
            assert sum(s.local_vocab[f] - 1 for s in specs) == v
        full = torch.arange(sum(vocab) * 2, dtype=torch.float32).view(-1, 2)
        shards = [s.shard_table(full) for s in specs]
        back = torch.zeros_like(full)
        specs[0].unshard_into(back, shards)
        assert torch.equal(back, full)
        # row i of field f sits at local index i // N + 1 of rank i % N
        o = 0
        for f, v in enumerate(vocab):
            for i in range(v):
                s = specs[i % N]
                lo = sum(s.local_vocab[:f])
                assert torch.equal(shards[i % N][lo + i // N + 1], full[o + i])
            o += v
        assert specs[0].bucket_capacity(4096) <= 4096
        # per-field buckets: never more slots than rows of the field an owner can hold, never more than the uniform bound
        bk = specs[0].buckets(4096)
        cap = specs[0].bucket_capacity(4096)
        assert bk.capf == [max(1, min(cap, -(-v // N))) for v in vocab]
        assert bk.S == sum(bk.capf) and bk.boff == [sum(bk.capf[:f]) for f in range(len(vocab))]
        sub_off, pay_base = bk.list_tables(N)
        F = len(vocab)
        for s_ in range(N):
            for f in range(F):
                assert int(sub_off[s_ * F + f]) == s_ * (bk.S + F) + bk.boff[f] + f + 1
                assert int(pay_base[s_ * F + f]) == s_ * bk.S + bk.boff[f]


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_exchange_protocol_gloo(world, tmp_path):
    codes, outs = run_ranks(world, ["cpu", str(tmp_path)])
    assert codes == [0] * world, "\n".join(outs)
    for r in range(world):
        assert os.path.exists(tmp_path / f"ok{r}")


@pytest.mark.parametrize("world", [2, 3])
def test_replica_exchange_gloo(world, tmp_path):
    """Replicated tables: the all-gathered (row, gradient) lists merge to the union batch's gradient on every rank."""
    codes, outs = run_ranks(world, ["cpu_replica", str(tmp_path)])
    assert codes == [0] * world, "\n".join(outs)
    for r in range(world):
        assert os.path.exists(tmp_path / f"ok{r}")
