"""Work of the NEXT batch done inside a step (FusedTrainer.step_tensors(..., next_idx=...)): the index sort of the
sparse backward riding in a backward GEMM launch (rs_gemm_f32_group_sort, a 256-thread form of the sort) and the replay
of skipped Adam steps per distinct row beside the update (rs_sparse_dense_update_ahead / rs_adam_catchup_plan).

Nothing here changes what a step computes (the reference's BaseModel.train_step, model/utils/BaseModel.py:141-150, with
its dense torch.optim.Adam): every test compares with the plain step -- which the golden traces and the oracle pin --
and asks for bit-identical weights, moments and losses."""
import pytest
import torch

pytestmark = pytest.mark.gpu

VOCAB_SMALL = [7, 40, 300, 2500, 20000, 3, 150000]


def build(kind, opt, vocab, batch, hidden=(64, 32), dim=16, seed=3, exact=True, graph=False):
    from recommendsystem_amd import data, models
    from recommendsystem_amd.base_model import random_all
    random_all(seed)
    rec = data.criteo_x1_recorder(embedding_dim=dim, batch_size=batch, vocab_sizes=list(vocab))
    mopt = "SGD" if opt == "Adagrad" else opt          # the model constructor knows the reference's three optimizers
    if kind == "deepfm":
        m = models.DeepFM(rec, 1e-2, 0, hidden_dim_list=list(hidden), normalization=None, optimizer_type=mopt)
    elif kind == "fm":
        m = models.FM(rec, 1e-2, 0, optimizer_type=mopt)
    else:
        m = models.DNN(rec, 1e-2, 0, hidden_dim_list=list(hidden), normalization=None, optimizer_type=mopt)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "embedded" in n:
                p[1:].normal_(0, 0.1)
    m.train()
    eng = m.use_fused_step(mode="sparse", use_graph=graph, exact_adam=exact, optimizer_type=opt)
    return m, eng


def state_of(m, eng):
    eng.flush()
    torch.cuda.synchronize()
    out = {k: v.detach().clone() for k, v in m.state_dict().items()}
    out["flat_w"] = eng.flat_w.clone()
    for i, t in enumerate(eng.tables):
        for k in ("m", "v"):
            if t[k] is not None:
                out[f"table{i}.{k}"] = t[k].clone()
    for k in ("flat_m", "flat_v"):
        if getattr(eng, k) is not None:
            out[k] = getattr(eng, k).clone()
    return out


def batches(vocab, batch, n, seed=11):
    from recommendsystem_amd import data
    idxs, labs = data.synthetic_batches(list(vocab), batch, n, "cpu", seed=seed)
    return [t.cuda() for t in idxs], [t.cuda() for t in labs]


def run(kind, opt, vocab, batch, order, announce, graph=False, exact=True, hidden=(64, 32), prepare=False):
    """order: batch numbers stepped on; announce[i]: the batch number announced at step i (None: nothing)."""
    m, eng = build(kind, opt, vocab, batch, hidden=hidden, exact=exact, graph=graph)
    idxs, labs = batches(vocab, batch, max(max(order), max(a for a in announce if a is not None) if any(
        a is not None for a in announce) else 0) + 1)
    if prepare:
        for i, b in enumerate(order):
            nxt = announce[i]
            eng.prepare(idxs[b], labs[b], next_idx=idxs[nxt] if nxt is not None else None,
                        ahead_slot=(i % 2) if i > 0 else None)
    losses = []
    for i, b in enumerate(order):
        nxt = announce[i]
        losses.append(eng.step_tensors(idxs[b], labs[b], resident=True,
                                       next_idx=idxs[nxt] if nxt is not None else None).clone())
    eng.check()
    return state_of(m, eng), torch.cat([l.view(-1) for l in losses]).cpu()


def assert_same(a, b, exact=True):
    sa, la = a
    sb, lb = b
    assert sa.keys() == sb.keys()
    if exact:
        assert torch.equal(la, lb), (la, lb)
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k
    else:
        torch.testing.assert_close(la, lb, rtol=1e-6, atol=0)
        for k in sa:
            torch.testing.assert_close(sa[k], sb[k], rtol=1e-5, atol=1e-8, msg=lambda s, k=k: f"{k}: {s}")


ORDER = [0, 1, 2, 0, 3, 3, 4, 1, 5, 2]


@pytest.mark.parametrize("kind,opt,exact", [
    ("deepfm", "Adam", True), ("deepfm", "AdamW", True), ("deepfm", "Adam", False), ("deepfm", "SGD", True),
    ("deepfm", "Adagrad", True), ("dnn", "Adam", True), ("fm", "Adam", True),
], ids=["deepfm-adam", "deepfm-adamw", "deepfm-lazy", "deepfm-sgd", "deepfm-adagrad", "dnn-adam", "fm-adam"])
@pytest.mark.parametrize("batch", [96, 1500])
def test_announced_batches_change_nothing(kind, opt, exact, batch):
    """Ten steps with rows that come back after one, two and several steps (and a batch stepped twice in a row):
    announcing every next batch gives the plain run's losses, weights and moments bit for bit."""
    plain = run(kind, opt, VOCAB_SMALL, batch, ORDER, [None] * len(ORDER), exact=exact)
    ahead = run(kind, opt, VOCAB_SMALL, batch, ORDER, ORDER[1:] + [None], exact=exact)
    assert_same(plain, ahead)


def test_broken_and_missing_promises():
    """A batch is announced and another one comes, announcements stop and start again: still the plain run -- to
    rounding: the rows of a batch that was announced and did not come have had part of their skipped steps replayed
    early, and a row's moments after k + n skipped steps are v * beta2^k * beta2^n against v * beta2^(k+n) (the
    reference multiplies by beta2 once per step; both forms stay within an ulp or two of it)."""
    plain = run("deepfm", "Adam", VOCAB_SMALL, 200, ORDER, [None] * len(ORDER))
    ann = [1, 5, None, 3, 3, 0, 1, None, 2, 4]          # steps 1, 4, 5, 7 and 9 get what was announced; 2 and 6 do not
    mixed = run("deepfm", "Adam", VOCAB_SMALL, 200, ORDER, ann)
    assert_same(plain, mixed, exact=False)
    # with SGD nothing is replayed: a broken promise leaves no trace at all
    plain = run("deepfm", "SGD", VOCAB_SMALL, 200, ORDER, [None] * len(ORDER))
    mixed = run("deepfm", "SGD", VOCAB_SMALL, 200, ORDER, ann)
    assert_same(plain, mixed)


@pytest.mark.parametrize("prepare", [False, True])
def test_announced_batches_under_graph_replay(prepare):
    """The same with every step replayed from a HIP graph (captured on first use, or all ahead of the first step)."""
    plain = run("deepfm", "Adam", VOCAB_SMALL, 300, ORDER, [None] * len(ORDER))
    order = list(range(6)) if prepare else ORDER         # prepared graphs: one per batch, slots alternate
    if prepare:
        plain = run("deepfm", "Adam", VOCAB_SMALL, 300, order, [None] * len(order))
    ahead = run("deepfm", "Adam", VOCAB_SMALL, 300, order, order[1:] + [None], graph=True, prepare=prepare)
    assert_same(plain, ahead)


def test_bench_shape_announced():
    """The benchmark's tower and batch (39 fields, 4096 samples, 624-1024-512-256-1): the sort rides beside the layer-1
    backward products."""
    from recommendsystem_amd import data
    vocab = [min(v, 50000) for v in data.criteo_x1_vocab_sizes()]
    order = [0, 1, 2, 3, 1]
    plain = run("deepfm", "Adam", vocab, 4096, order, [None] * 5, hidden=(1024, 512, 256))
    ahead = run("deepfm", "Adam", vocab, 4096, order, order[1:] + [None], hidden=(1024, 512, 256))
    assert_same(plain, ahead)


@pytest.mark.parametrize("halves", [False, True], ids=["whole", "two-halves"])
@pytest.mark.parametrize("B", [1, 63, 64, 1000, 1024, 1025, 4000, 4096])
def test_sort_in_gemm_launch_writes_the_same_plan(B, halves):
    """rs_gemm_f32_group_sort (here without products: the sort alone, 256 threads per field; whole, or cut in the two
    launches it rides in) against rs_sparse_bwd_sort."""
    from recommendsystem_amd import ops
    vocab = [2, 5, 64, 1000, 70000, 1000000, 33]
    layout = ops.FieldLayout(vocab, "cuda")
    g = torch.Generator().manual_seed(B)
    idx = torch.stack([torch.randint(0, v, (B,), generator=g) for v in vocab]).cuda()
    idx[3, : B // 2] = 0                                   # padding entries
    a = ops.SparsePlan(layout, cap=B, d=16, with_lr=True)
    b = ops.SparsePlan(layout, cap=B, d=16, with_lr=True)
    a.sort_columnar(idx)
    if halves:
        b._sort_in_group(None, 0, idx, 1)
        b._sort_in_group(None, 0, idx, 2)
    else:
        b._sort_in_group(None, 0, idx)
    torch.cuda.synchronize()
    n_plan = len(vocab) * (6 * B + 2 + 4)                  # everything but the two-level sort's scratch
    ca, cb = a.counts().cpu(), b.counts().cpu()
    assert torch.equal(ca, cb)
    F = len(vocab)
    for f in range(F):
        n, nch, nseg = (int(x) for x in ca[f, :3])
        for name, off, per, used in (("sorted_src", 0, B, n), ("entry_chunk", F * B, B, n), ("entry_seg", 2 * F * B, B, n),
                                     ("chunk_start", 3 * F * B, B + 1, nch + 1),
                                     ("seg_row", 3 * F * B + F * (B + 1), B, nseg),
                                     ("seg_first_chunk", 4 * F * B + F * (B + 1), B + 1, nseg + 1)):
            x = a.plan[off + f * per: off + f * per + used]
            y = b.plan[off + f * per: off + f * per + used]
            assert torch.equal(x, y), (name, f)
    assert n_plan <= a.plan.numel()
