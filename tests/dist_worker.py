"""One rank of the multi-process tests (started by test_dist_cpu.py / test_dist_gpu.py).

    python dist_worker.py cpu <outdir>     host-side exchange protocol under gloo with torch-CPU stand-in kernels
    python dist_worker.py gpu <outdir> <kind> <optimizer> [onestream|generic|replicated]
                                           DistributedTrainer (generic: DistributedSparseTrainer; replicated:
                                           ReplicatedTablesTrainer) on cuda:0 (all ranks share the one GPU of the test
                                           box), gloo with host staging in place of RCCL
    python dist_worker.py gpu1 <outdir> <kind> <optimizer>
                                           one rank with the real RCCL backend: eager, whole-step HIP graph, graph with
                                           the second stream, replicated tables

RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from the environment.
"""
import faulthandler
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

VOCAB = [7, 50, 3, 400, 1000, 23]
B, D = 96, 8


def rank_batch(rank, vocab=VOCAB, batch=B, zero_every=11):
    g = torch.Generator().manual_seed(100 + rank)
    idx = torch.stack([torch.randint(1, v, (batch,), generator=g) for v in vocab], 0)
    idx[:, ::zero_every] = 0                              # padding entries
    idx[1, :] = idx[1, 0].item() if idx[1, 0] > 0 else 5  # one field with a single hot key
    labels = (torch.rand(batch, generator=g) < 0.3).long()
    return idx, labels


def full_tables(vocab=VOCAB, d=D):
    g = torch.Generator().manual_seed(7)
    R = sum(vocab)
    t = torch.randn(R, d, generator=g)
    t1 = torch.randn(R, generator=g)
    o = 0
    for v in vocab:
        t[o] = 0
        t1[o] = 0
        o += v
    return t, t1


def offsets(vocab):
    off, o = [], 0
    for v in vocab:
        off.append(o)
        o += v
    return torch.tensor(off).view(-1, 1)


def run_cpu(outdir):
    from recommendsystem_amd.dist import Comm, ShardSpec, ShardedExchange
    from shard_cpu_kernels import TorchShardKernels
    rank, N = dist.get_rank(), dist.get_world_size()
    comm = Comm()
    spec = ShardSpec(VOCAB, N, rank)
    F = len(VOCAB)
    T, T1 = full_tables()
    shard, shard1 = spec.shard_table(T), spec.shard_table(T1.view(-1, 1)).view(-1)
    assert shard.shape[0] == sum(spec.local_vocab)

    # shard <-> full round trip through all_gather
    mx = torch.tensor([shard.shape[0]])
    sizes = [int(x) for x in comm.all_gather(mx)]
    pad = torch.zeros(max(sizes), D)
    pad[:shard.shape[0]] = shard
    back = torch.zeros_like(T)
    spec.unshard_into(back, [g[:k] for g, k in zip(comm.all_gather(pad), sizes)])
    assert torch.equal(back, T)

    idx, _ = rank_batch(rank)
    bk = spec.buckets(B)                       # per-field capacities: small fields get few slots
    assert min(bk.capf) < max(bk.capf) or len(set(VOCAB)) == 1
    k = TorchShardKernels(spec, B, D, True, True, bk)
    ex = ShardedExchange(k, comm, spec, B, D, True, True, bk, "cpu")
    emb, lr1 = torch.full((B * F, D), 9.0), torch.full((B * F,), 9.0)
    ex.forward(idx, shard, shard1, emb, lr1)
    flat = (idx + offsets(VOCAB)).t().reshape(-1)                  # entry b*F+f
    assert torch.equal(emb, T[flat]), "gathered rows differ"
    assert torch.equal(lr1, T1[flat])
    ex.check_overflow()

    # backward: the owners' merged gradients equal the global dense gradient restricted to their rows
    g = torch.Generator().manual_seed(500 + rank)
    gout = torch.randn(B * F, D, generator=g)
    dlr = torch.randn(B, generator=g)
    part = torch.zeros(k.n_own_partials)
    ex.backward(gout, None, None, None, dlr, part)
    all_idx = comm.all_gather(idx)
    all_g = comm.all_gather(gout)
    all_d = comm.all_gather(dlr)
    G, G1 = torch.zeros_like(T), torch.zeros_like(T1)
    for r in range(N):
        fl = (all_idx[r] + offsets(VOCAB)).t().reshape(-1)
        keep = (all_idx[r].t().reshape(-1) > 0)
        G.index_add_(0, fl[keep], all_g[r][keep])
        G1.index_add_(0, fl[keep], all_d[r].repeat_interleave(F)[keep])
    # (sums of up to N*B terms per row, added in a different order than index_add_'s)
    torch.testing.assert_close(k.own_grad, spec.shard_table(G), rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(k.own_grad1, spec.shard_table(G1.view(-1, 1)).view(-1), rtol=1e-4, atol=2e-5)
    sq = part[:1].clone()
    comm.all_reduce(sq)
    torch.testing.assert_close(sq[0], (G ** 2).sum() + (G1 ** 2).sum(), rtol=1e-5, atol=0)

    # a bucket that is too small is reported, and the collectives still line up
    from recommendsystem_amd.dist import BucketLayout
    small = BucketLayout([min(4, c) for c in bk.capf])
    k2 = TorchShardKernels(spec, B, D, True, False, small)
    ex2 = ShardedExchange(k2, comm, spec, B, D, True, False, small, "cpu")
    ex2.forward(idx, shard, None, emb, None)
    flag = ex2.overflow.clone()
    comm.all_reduce(flag)
    assert int(flag) > 0
    if int(ex2.overflow) != 0:
        try:
            ex2.check_overflow()
        except RuntimeError as e:
            assert "overflow" in str(e)
        else:
            raise AssertionError("overflow not reported")
    open(os.path.join(outdir, f"ok{rank}"), "w").write("ok")


def run_cpu_replica(outdir):
    """ReplicaExchange (all-gather of every rank's distinct rows and gradients, merge in rank order) under gloo."""
    from recommendsystem_amd.dist import Comm, ReplicaExchange
    from shard_cpu_kernels import TorchReplicaKernels
    rank, N = dist.get_rank(), dist.get_world_size()
    comm = Comm()
    F = len(VOCAB)
    off = offsets(VOCAB).view(-1)

    def local(r):
        idx, _ = rank_batch(r)
        g = torch.Generator().manual_seed(900 + r)
        seg_rows = torch.full((F, B), -7, dtype=torch.int32)           # garbage behind the valid prefix
        n_seg = torch.zeros(F, dtype=torch.int32)
        for f in range(F):
            u = torch.unique(idx[f][idx[f] > 0])
            seg_rows[f, :len(u)] = (u + off[f]).int()
            n_seg[f] = len(u)
        return seg_rows, n_seg, torch.randn(F, B, D, generator=g), torch.randn(F, B, generator=g)

    seg_rows, n_seg, rg, rg1 = local(rank)
    k = TorchReplicaKernels(VOCAB, N, B, D)
    rx = ReplicaExchange(k, comm, F, B, D, True, True, "cpu")
    rx.merge(seg_rows, n_seg, rg, rg1)
    G, G1 = torch.zeros(sum(VOCAB), D), torch.zeros(sum(VOCAB))
    for r in range(N):                                                  # every rank can rebuild every rank's lists
        sr, ns, a, a1 = local(r)
        for f in range(F):
            n = int(ns[f])
            G.index_add_(0, sr[f, :n].long(), a[f, :n])
            G1.index_add_(0, sr[f, :n].long(), a1[f, :n])
    assert torch.equal(k.grad, G) and torch.equal(k.grad1, G1), "merged gradients differ"
    torch.testing.assert_close(rx.partials[0], (G ** 2).sum() + (G1 ** 2).sum(), rtol=1e-6, atol=0)
    # every rank holds the same result bit for bit (the replicas must not drift)
    for other in comm.all_gather(k.grad):
        assert torch.equal(other, k.grad)
    open(os.path.join(outdir, f"ok{rank}"), "w").write("ok")


GPU_VOCAB = [7, 50, 3, 400, 1000, 23, 5000, 64]
GPU_BATCH, GPU_DIM, GPU_STEPS = 256, 8, 3


def build_gpu_model(kind, optimizer, batch):
    """The model of the multi-process GPU tests (same seed on every rank and in the single-GPU reference)."""
    from recommendsystem_amd import data, models
    from recommendsystem_amd.base_model import random_all
    vocab = GPU_VOCAB
    random_all(11)
    rec = data.criteo_x1_recorder(embedding_dim=GPU_DIM, batch_size=batch, vocab_sizes=vocab)
    kw = dict(optimizer_type=optimizer if optimizer != "Adagrad" else "SGD")
    # "<kind>_bn": the reference's default towers (BatchNorm in every block, Layers.py:9)
    norm = "bn" if kind.endswith("_bn") else None
    kind = kind[:-3] if norm else kind
    hid = dict(hidden_dim_list=[64, 32], normalization=norm)
    if kind == "deepfm":
        m = models.DeepFM(rec, 1e-2, 0, **hid, **kw)
    elif kind == "fm":
        m = models.FM(rec, 1e-2, 0, **kw)
    elif kind == "lr":
        m = models.LR(rec, 1e-2, 0, **kw)
    elif kind == "dnn":
        m = models.DNN(rec, 1e-2, 0, **hid, **kw)
    elif kind == "pnn":
        m = models.PNN(rec, 1e-2, 0, pnn_type="inner", **hid, **kw)
    elif kind == "dcn":
        m = models.DCN(rec, 1e-2, 0, cross_layers_num=2, **hid, **kw)
    elif kind == "nfm":
        m = models.NFM(rec, 1e-2, 0, hidden_dim=4, **hid, **kw)
    elif kind == "afm":
        m = models.AFM(rec, 1e-2, 0, attention_dim=4, **kw)
    elif kind == "ffm":
        m = models.FFM(rec, 1e-2, 0, **kw)
    else:
        raise ValueError(kind)
    # larger initial rows than the reference's 1e-4 so that every term of the gradient matters
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 2 and p.shape[0] in vocab and ("embedded" in n or "tables" in n):
                p[1:].normal_(0, 0.05)
            if n == "v" or n.startswith("cross."):
                p.mul_(0.2)
    m.train()
    return m


def run_gpu1(outdir, kind, optimizer):
    """One rank, RCCL: the same three steps eagerly, from one HIP graph, from that graph with the second stream, and
    with replicated tables from a graph."""
    from recommendsystem_amd.dist import DistributedTrainer, ReplicatedTablesTrainer
    torch.cuda.set_device(0)
    vocab, batch, steps = GPU_VOCAB, GPU_BATCH, GPU_STEPS
    data = [rank_batch(10 * s, vocab, batch, zero_every=17) for s in range(steps + 1)]
    data = [(i.cuda(), y.cuda()) for i, y in data]
    for name, cls, kw in (("eager", DistributedTrainer, dict(use_graph=False, overlap_comm=False)),
                          ("graph", DistributedTrainer, dict(use_graph=True, overlap_comm=False)),
                          ("graph_overlap", DistributedTrainer, dict(use_graph=True, overlap_comm=True)),
                          ("eager_ahead", DistributedTrainer, dict(use_graph=False, overlap_comm=True)),
                          ("graph_ahead", DistributedTrainer, dict(use_graph=True, overlap_comm=True)),
                          ("replicated_graph", ReplicatedTablesTrainer, dict(use_graph=True))):
        print("variant", name, flush=True)
        m = build_gpu_model(kind, optimizer, batch)
        eng = cls(m, max_norm=0.05, optimizer_type=optimizer, **kw)
        losses = []
        for s in range(steps):
            idx, y = data[s]
            if name.endswith("_ahead"):       # every next batch announced (the last announcement is not kept)
                loss = eng.step_tensors(idx, y, resident=True, next_idx=data[s + 1][0])
            else:
                loss = eng.step_tensors(idx, y)
            losses.append(float(loss.item()))
        eng.check()
        eng.sync_to_model()
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        sd["__losses__"] = np.array(losses)
        np.savez(os.path.join(outdir, f"state_{name}.npz"), **sd)
        # graphs that captured collectives go before the communicator does
        eng.release_graphs() if hasattr(eng, "release_graphs") else getattr(eng, "_graphs", {}).clear()
        del eng, m
        import gc
        gc.collect()
        torch.cuda.synchronize()
    open(os.path.join(outdir, "ok0"), "w").write("ok")


def run_gpu(outdir, kind, optimizer, generic=False, replicated=False, overlap_comm=True):
    from recommendsystem_amd.dist import DistributedSparseTrainer, DistributedTrainer, ReplicatedTablesTrainer
    rank, N = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    vocab, batch, steps = GPU_VOCAB, GPU_BATCH, GPU_STEPS
    m = build_gpu_model(kind, optimizer, batch)
    if generic:
        eng = DistributedSparseTrainer(m, max_norm=0.05, stage_cpu=True, optimizer_type=optimizer)
    elif replicated:
        eng = ReplicatedTablesTrainer(m, max_norm=0.05, stage_cpu=True, optimizer_type=optimizer)
    else:
        eng = DistributedTrainer(m, max_norm=0.05, stage_cpu=True, optimizer_type=optimizer, overlap_comm=overlap_comm)
    losses = []
    for s in range(steps):
        idx, y = rank_batch(rank + 10 * s, vocab, batch, zero_every=17)
        loss = eng.step_tensors(idx.cuda(), y.cuda())
        losses.append(float(loss.item()))
    eng.check()
    eng.sync_to_model()
    if rank == 0:
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        np.savez(os.path.join(outdir, "state.npz"), **sd)
    np.save(os.path.join(outdir, f"loss{rank}.npy"), np.array(losses))
    open(os.path.join(outdir, f"ok{rank}"), "w").write("ok")


if __name__ == "__main__":
    faulthandler.enable()               # a crash in native code still names the Python frame it came from
    mode, outdir = sys.argv[1], sys.argv[2]
    if mode == "gpu1":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    try:
        if mode == "cpu":
            run_cpu(outdir)
        elif mode == "cpu_replica":
            run_cpu_replica(outdir)
        elif mode == "gpu1":
            run_gpu1(outdir, sys.argv[3], sys.argv[4])
        else:
            extra = sys.argv[5] if len(sys.argv) > 5 else ""
            run_gpu(outdir, sys.argv[3], sys.argv[4], generic=extra == "generic", replicated=extra.startswith("replicated"),
                    overlap_comm=extra != "onestream")
    finally:
        dist.destroy_process_group()
