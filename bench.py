#!/usr/bin/env python3
"""Headline benchmark: DeepFM training step on synthetic Criteo_x1-shaped batches.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (RANK/LOCAL_RANK/WORLD_SIZE from the environment when N > 1; started without them, e.g. as plain
`python bench.py --gpus 8`, the script starts the N rank processes itself before anything touches a GPU).  A "step" is one
pass of the whole hot path over one batch of B samples per GPU: fused gather + LR + FM, the MLP
GEMMs forward and backward, the loss head, the sparse embedding backward, global-norm clipping and
the optimizer update.  Inputs are resident in HBM before the timed region starts.  Rank 0 prints
ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (spec)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--batch", type=int, default=4096, help="samples per GPU per step")
    p.add_argument("--dim", type=int, default=16)
    p.add_argument("--hidden", type=str, default="1024,512,256")
    p.add_argument("--model", type=str, default="deepfm",
                   choices=["deepfm", "fm", "lr", "dnn", "widedeep", "ffm", "pnn", "pnn_inner", "dcn", "nfm", "afm"],
                   help="deepfm is the headline; ffm..afm run through the generic touched-rows trainer")
    p.add_argument("--fields", type=int, default=39, choices=[39, 26],
                   help="39 = 13 small-vocabulary + 26 categorical fields; 26 = the categorical fields only")
    p.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = min(16, cores))")
    p.add_argument("--optimizer", type=str, default="Adam")
    p.add_argument("--age-steps", type=int, default=0,
                   help="untimed training steps on further non-repeating batches before the warm-up (single GPU): exact "
                        "Adam replays a returning row's skipped steps, so the step costs more once the tables have a "
                        "history; some thousand steps reach the level a long run settles at")
    p.add_argument("--regular", type=float, default=0.0,
                   help="emb_regular = net_regular (BaseModel.py:214-231); with --optimizer SGD the touched-rows step "
                        "carries the tables' shrink in one scalar, other optimizers need --mode dense; not part of "
                        "the CPU baseline (use --cpu-baseline 0)")
    p.add_argument("--activation", type=str, default="ReLU",
                   help="activation of the MLP blocks (the reference's default ReLU also sits on the 1-wide last block)")
    p.add_argument("--dropout", type=float, default=0.0,
                   help="dropout_rate of the first MLP block (the reference's scripts pass one; 0 = none)")
    p.add_argument("--mode", type=str, default="sparse", choices=["sparse", "dense"])
    p.add_argument("--exact-adam", type=int, default=1,
                   help="touched-rows Adam/AdamW with the reference's dense semantics (a row's skipped zero-gradient steps "
                        "are replayed before it is read); 0 = lazy Adam (absent rows keep weights and moments)")
    p.add_argument("--launch-check", type=int, default=0,
                   help="N>1: only prove the launch plumbing (rendezvous over gloo, one all-reduce, rank 0 prints a JSON "
                        "line); touches no GPU")
    p.add_argument("--tables", type=str, default="sharded", choices=["sharded", "replicated"],
                   help="multi-GPU placement of the embedding tables: row-sharded with an all-to-all exchange, or a "
                        "replica per rank with an all-gather of (row, gradient) pairs")
    p.add_argument("--graph", type=int, default=1, help="replay the step from a HIP graph")
    p.add_argument("--overlap", type=int, default=0, help="run weight-gradient GEMMs and the index sort on side streams")
    p.add_argument("--uniform", type=int, default=0, help="uniform instead of Zipf(1.1) indices")
    p.add_argument("--cpu-baseline", type=int, default=1)
    p.add_argument("--cpu-seconds", type=float, default=20.0)
    p.add_argument("--distinct-batches", type=int, default=0,
                   help="resident synthetic batches the steps cycle through; 0 (default) = warmup + steps, i.e. no batch "
                        "is seen twice, so every step meets rows that are as stale as a first pass over the data makes them")
    p.add_argument("--ahead", type=int, default=-1,
                   help="1 = announce every next batch to the step before it (step_tensors(..., next_idx=...)), as a feeder "
                        "that stages batches ahead allows; 0 = every step prepares its own batch in front of its lookup.  "
                        "Several GPUs (default 1): the key exchange of the next batch -- sort, bucketing, all-to-all -- runs "
                        "on the second stream beside the step.  One GPU (default 0, measured faster): the index sort rides "
                        "in two backward GEMM launches -- where it runs three to five times as long as alone -- and the Adam "
                        "catch-up beside the update")
    p.add_argument("--force-dist", type=int, default=0,
                   help="N=1 only: run the row-sharded multi-GPU trainer with a one-rank RCCL group (what the sharded "
                        "path costs on top of the single-GPU step, and a check of the RCCL calls on one GPU)")
    p.add_argument("--dist-graph", type=int, default=1,
                   help="N>1: 1 (default) = the whole step, RCCL collectives included, is one HIP graph per resident batch; "
                        "0 = issued eagerly (host-bound: about thirty entry-point calls and five collectives per step)")
    p.add_argument("--overlap-comm", type=int, default=-1,
                   help="sharded tables: all-reduce of the dense gradient on a second stream and communicator beside the "
                        "gradient all-to-all and the owner-side reduction, and the key exchange of the next batch on a "
                        "third (default); 0: everything on one stream")
    p.add_argument("--event-pass", type=int, default=1,
                   help="after the timed region, run the same K steps eagerly with HIP events around every launch "
                        "(per-kernel durations for the roofline figures)")
    p.add_argument("--check-every", type=int, default=0,
                   help="steps between two looks at the device-side error flags (0: the trainer's default)")
    p.add_argument("--share-launches", type=int, default=1,
                   help="0: kernels that share a launch in the step (sort + lookup, reduction + dense norm, the two optimizer "
                        "steps) are launched separately in the timed region too -- for counter passes, which attribute "
                        "their readings per kernel")
    p.add_argument("--rehearse-on-one-gpu", type=int, default=0,
                   help="N>1 only: all ranks share cuda:0 and exchange through gloo with host staging (functional "
                        "rehearsal of the multi-GPU path on a one-GPU box; the numbers mean nothing)")
    return p.parse_args()


def build(args, device):
    from recommendsystem_amd import data, models
    from recommendsystem_amd.base_model import random_all
    random_all(2021)
    vs = data.criteo_x1_vocab_sizes()
    if args.fields == 26:
        vs = vs[13:]
    rec = data.criteo_x1_recorder(embedding_dim=args.dim, batch_size=args.batch, vocab_sizes=vs)
    hidden = [int(h) for h in args.hidden.split(",") if h]
    gpu = device.index
    kw = dict(optimizer_type=args.optimizer)
    if args.regular > 0:
        kw["regular"] = args.regular
    if args.model in ("deepfm", "dnn", "widedeep", "pnn", "pnn_inner", "dcn", "nfm"):
        kw["activation_func"] = args.activation
        if args.dropout > 0:
            kw["dropout_rate"] = args.dropout
    if args.model == "deepfm":
        m = models.DeepFM(rec, 1e-3, gpu, hidden_dim_list=hidden, normalization=None, **kw)
    elif args.model == "fm":
        m = models.FM(rec, 1e-3, gpu, **kw)
    elif args.model == "lr":
        m = models.LR(rec, 1e-3, gpu, **kw)
    elif args.model == "dnn":
        m = models.DNN(rec, 1e-3, gpu, hidden_dim_list=hidden, normalization=None, **kw)
    elif args.model == "widedeep":
        m = models.WideDeep(rec, 1e-3, gpu, hidden_dim_list=hidden, normalization=None, **kw)
    elif args.model == "ffm":
        m = models.FFM(rec, 1e-3, gpu, **kw)
    elif args.model in ("pnn", "pnn_inner"):
        m = models.PNN(rec, 1e-3, gpu, hidden_dim_list=hidden, pnn_type="both" if args.model == "pnn" else "inner",
                       normalization=None, **kw)
    elif args.model == "dcn":
        m = models.DCN(rec, 1e-3, gpu, hidden_dim_list=hidden, normalization=None, **kw)
    elif args.model == "nfm":
        m = models.NFM(rec, 1e-3, gpu, hidden_dim_list=hidden, normalization=None, **kw)
    else:
        m = models.AFM(rec, 1e-3, gpu, **kw)
    m.train()
    return m, rec, hidden


def cpu_baseline(args, rec, hidden, idx_cpu, lab_cpu):
    """The CPU oracle's training step (oracle/ctr_oracle.py: per-field nn.Embedding, torch.stack,
    sum-square, nn.Linear stack, BCE, clip_grad_norm_, dense torch optimizer) on the same batches."""
    from oracle import ctr_oracle as O
    cores = args.cpu_threads or min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    torch.manual_seed(2021)
    okw = {"ffm": {}, "afm": {}, "pnn": {"pnn_type": "both"}, "pnn_inner": {"pnn_type": "inner"}}.get(
        args.model, {"hidden": hidden, "normalization": None})
    if args.model in ("pnn", "pnn_inner"):
        okw.update(hidden=hidden, normalization=None)
    if "hidden" in okw:
        okw["activation"] = args.activation
    m = O.OracleCTRModel(args.model.split("_")[0], rec.feature_map, rec.features, args.dim, **okw)
    m.train()
    opt = O.make_optimizer(args.optimizer, m.parameters(), 1e-3)
    cols = rec.features

    def one(i):
        idx = idx_cpu[i % len(idx_cpu)]
        x = {c: idx[j] for j, c in enumerate(cols)}
        O.train_step(m, opt, x, lab_cpu[i % len(lab_cpu)])

    n_warm = 10
    for i in range(n_warm):
        one(i)
    t0 = time.perf_counter()
    n = 0
    while True:
        one(n + n_warm)
        n += 1
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds or n >= 50:
            break
    return {"value": n * args.batch / el, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of batch {args.batch} after {n_warm} warm-up steps, same synthetic batches, "
                      f"torch CPU ops with {cores} threads ({el:.1f} s)"}


class _TwoSlotFeeder:
    """Batch i is served from input buffer pair i % 2; a side stream copies batch i + 2 into the pair as soon as step i
    is through with it, i.e. while step i + 1 runs -- the device-side picture of a feeder that stages batches ahead.
    With a trainer that exchanges the keys of an announced batch on a stream of its own (`stream`), the copies go
    through that stream: the key exchange of batch i + 1 then follows its copy in stream order, and the step, which
    waits for the key exchange, needs no event of the feeder's."""

    def __init__(self, idxs, labs, device, stream=None):
        self.idxs, self.labs = idxs, labs
        self.n = len(idxs)
        self.buf = [(idxs[k % self.n].clone(), labs[k % self.n].clone()) for k in range(2)]
        self.shared = stream is not None
        self.side = stream if stream is not None else torch.cuda.Stream(device=device)
        self.filled = [None, None]          # event: the copy into pair k has finished

    def take(self, i):
        k = i % 2
        if not self.shared and self.filled[k] is not None:
            torch.cuda.current_stream().wait_event(self.filled[k])
        return self.buf[k][0], self.buf[k][1], self.buf[(i + 1) % 2][0]

    def done(self, i):
        """Step i has been issued: refill its buffer pair with batch i + 2 once it has run."""
        k = i % 2
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            self.buf[k][0].copy_(self.idxs[(i + 2) % self.n])
            self.buf[k][1].copy_(self.labs[(i + 2) % self.n])
            if not self.shared:
                self.filled[k] = torch.cuda.Event()
                self.filled[k].record()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (one per GPU, rendezvous on a free
    local port) and exit with their status.  Runs before anything in this process touches a GPU."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in live:               # a rank died: the others would wait in a collective forever
                    q.terminate()
        time.sleep(0.05)
    raise SystemExit(rc)


def launch_check(rank, world):
    """Rendezvous and one collective over gloo; no GPU involved."""
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([1 << rank], dtype=torch.int64)
    dist.all_reduce(t)
    ok = int(t.item()) == (1 << world) - 1
    if rank == 0:
        print(json.dumps({"launch_check": "ok" if ok else "FAILED", "world": world, "rank_mask": int(t.item())}), flush=True)
    dist.destroy_process_group()
    raise SystemExit(0 if ok else 1)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1")) if args.gpus > 1 else 1
    rank = int(os.environ.get("RANK", "0")) if args.gpus > 1 else 0
    local = int(os.environ.get("LOCAL_RANK", "0")) if args.gpus > 1 else 0
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly --gpus ranks")
    if args.launch_check and world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        launch_check(rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    rehearse = bool(args.rehearse_on_one_gpu) and world > 1
    device = torch.device("cuda", 0 if rehearse else local)
    torch.cuda.set_device(device)
    use_dist = world > 1 or bool(args.force_dist)
    stdout_fd = None
    if use_dist:
        # RCCL prints a version banner (and gloo its connection lines) on the C-level stdout when a communicator comes
        # up: everything but the result line goes to stderr, so that stdout carries the ONE JSON line and nothing else
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from recommendsystem_amd import data, ops
    from recommendsystem_amd.prof import KernelTimer

    model, rec, hidden = build(args, device)
    vocab = [rec.feature_map[c]["vocab_size"] for c in rec.features]
    kind = getattr(model, "KIND", None)
    fused_kind = kind in ("lr", "fm", "deepfm", "dnn", "widedeep", "dcn") or (kind == "pnn" and model.pnn_type == "inner")
    if use_dist and fused_kind:
        from recommendsystem_amd.dist import DistributedTrainer, ReplicatedTablesTrainer
        cls = ReplicatedTablesTrainer if args.tables == "replicated" else DistributedTrainer
        kw = {"overlap_comm": None if args.overlap_comm < 0 else bool(args.overlap_comm)} if cls is DistributedTrainer else {}
        # gloo with host staging (the one-GPU rehearsal) cannot be captured
        eng = cls(model, mode=args.mode, use_graph=bool(args.dist_graph) and not rehearse, stage_cpu=rehearse, **kw)
    elif use_dist:
        from recommendsystem_amd.dist import DistributedSparseTrainer
        eng = DistributedSparseTrainer(model, stage_cpu=rehearse)
    elif fused_kind:
        eng = model.use_fused_step(mode=args.mode, use_graph=bool(args.graph), overlap=int(args.overlap),
                                   exact_adam=bool(args.exact_adam))
    else:
        eng = model.use_sparse_step(use_graph=bool(args.graph), exact_adam=bool(args.exact_adam))

    if not args.share_launches and getattr(eng, "_fuse_small", False):
        eng._fuse_small = False
    if args.check_every > 0:
        eng.check_every = args.check_every

    nb = args.distinct_batches if args.distinct_batches > 0 else args.warmup + args.steps
    idxs, labs = data.synthetic_batches(vocab, args.batch, nb, "cpu", seed=2021 + rank, uniform=bool(args.uniform))
    idx_dev = [t.to(device) for t in idxs]
    lab_dev = [t.to(device) for t in labs]

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()

    # The pre-staged batches stay resident: with --graph one graph per batch buffer, replayed without copies.  The
    # graphs are captured first (capturing leaves the training state as it was), so that the W warm-up steps and the K
    # timed steps walk through W + K different batches when that many are resident (the default).
    use_next = args.ahead == 1 and hasattr(eng, "_ahead_supported") and eng._ahead_supported(args.batch)

    def announce(i):
        return {"next_idx": idx_dev[(i + 1) % nb]} if use_next else {}

    feeder = None
    if use_dist:
        # Several GPUs: the step reads its batch from one of TWO input buffer pairs, which a side stream fills one step
        # ahead (what a feeder's host-to-device copy does while the step before runs): two graphs with the RCCL
        # collectives inside instead of one per batch, and the step can be told its next batch (--ahead).
        use_next = args.ahead != 0 and getattr(eng, "_ahead_kind", "") == "keys" and eng._ahead_supported(args.batch)
        feeder = _TwoSlotFeeder(idx_dev, lab_dev, device, stream=eng.side_keys if (use_next and eng.use_graph) else None)

        def run_step(i):
            idx, lab, nxt = feeder.take(i)
            kw = {"next_idx": nxt} if use_next else {}
            out = eng.step_tensors(idx, lab, resident=True, **kw) if hasattr(eng, "_graph_key") else eng.step_tensors(idx, lab)
            feeder.done(i)
            return out
    else:
        def run_step(i):
            return eng.step_tensors(idx_dev[i % nb], lab_dev[i % nb], resident=True, **announce(i))
        if eng.use_graph and hasattr(eng, "prepare"):
            for i in range(args.warmup + args.steps):
                kw = dict(announce(i), ahead_slot=(i % 2) if i > 0 else None) if use_next else {}
                eng.prepare(idx_dev[i % nb], lab_dev[i % nb], **kw)
    aged = 0
    if args.age_steps > 0 and not use_dist:
        while aged < args.age_steps:
            n = min(64, args.age_steps - aged)
            bi, bl = data.synthetic_batches(vocab, args.batch, n, device, seed=77_000 + aged, uniform=bool(args.uniform))
            for i in range(n):
                eng.step_tensors(bi[i], bl[i])       # (with --graph: one graph over static buffers the batch is copied into)
            aged += n
        torch.cuda.synchronize()
    for i in range(args.warmup):
        run_step(i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()

    step_events = [] if os.environ.get("RS_BENCH_STEP_TIMES") else None      # diagnostic: GPU time of every timed step
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        loss = run_step(i)
        if step_events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            step_events.append((ev, time.perf_counter() - t0))
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if step_events:
        gpu = [round(a[0].elapsed_time(b[0]) * 1e3) for a, b in zip(step_events, step_events[1:])]
        host = [round((b[1] - a[1]) * 1e6) for a, b in zip(step_events, step_events[1:])]
        print("step GPU us:", gpu, file=sys.stderr)
        print("step host issue us:", host, file=sys.stderr)

    # Per-launch durations: the same K steps issued eagerly with a HIP event pair around every launch, on the
    # launch stream.  Matrix products in front of every step keep the GPU busy while the host enqueues the step, so
    # that each event pair brackets only its kernel (no host launch latency between the two timestamps).
    timer = KernelTimer(names=None)
    measured_in = "separate eager pass of the same K steps, HIP events around every launch"
    if args.event_pass:
        was_graph = eng.use_graph
        eng.use_graph = False
        # time every kernel in a launch of its own: the timed region shares launches between independent kernels
        # (sort + gather, reduction + dense norm, the two optimizer steps), which would hide the gather's duration
        was_fused = getattr(eng, "_fuse_small", None)
        if was_fused:
            eng._fuse_small = False
        # The cover in front of every step is matrix work, not a one-wave spin kernel: a GPU that has idled for
        # milliseconds runs its next launches at lower clocks than the back-to-back replays of the timed region do.
        cover_a = torch.randn(4096, 4096, device=device)
        cover_b = torch.empty_like(cover_a)

        def cover(ms):
            for _ in range(max(1, int(round(ms / cover.ms_each)))):
                torch.mm(cover_a, cover_a, out=cover_b)
        torch.mm(cover_a, cover_a, out=cover_b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            torch.mm(cover_a, cover_a, out=cover_b)
        e1.record(); torch.cuda.synchronize()
        cover.ms_each = max(e0.elapsed_time(e1) / 4, 1e-2)
        for i in range(2):
            eng.step_tensors(idx_dev[i % nb], lab_dev[i % nb])
        torch.cuda.synchronize()
        timer.calibrate()
        with timer:
            for i in range(args.steps):
                cover(6.0)
                eng.step_tensors(idx_dev[i % nb], lab_dev[i % nb])
                torch.cuda.synchronize()
        eng.use_graph = was_graph
        if was_fused:
            eng._fuse_small = True
            measured_in += "; kernels that share a launch in the timed region are launched separately here"
    summ = timer.summary()
    final_loss = float(loss.item())
    exchange_check = None
    if use_dist:
        try:
            eng.check()      # bucket overflow / index range flags of the exchange (device sync, outside the timing)
            exchange_check = "ok"
        except Exception as e:       # reported with the result instead of losing the line
            exchange_check = f"FAILED: {e}"

    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel family (the fp32 MFMA GEMM) -------------------------------
    kernels = []
    gemm_flop = gemm_us = 0.0
    gather = None
    per_step = {}
    by_layout = {}
    for (name, shape), a in sorted(summ.items(), key=lambda kv: -kv[1]["total_us"]):
        per_step[name] = per_step.get(name, 0.0) + a["total_us"] / args.steps
        ent = {"entry": name, "shape": shape, "launches": a["n"], "avg_us": round(a["avg_us"], 2)}
        if a["unit"] == "flop":
            ent["tflops"] = round(a["work"] / a["avg_us"] / 1e6, 2)
            ent["frac_of_peak"] = round(a["work"] / a["avg_us"] / 1e6 / MFMA_F32_PEAK_TFLOPS, 4)
            gemm_flop += a["work"] * a["n"]
            gemm_us += a["total_us"]
            # the kernels a profiler lists: NT = forward <false,false>, NN = dX <false,true> and TN = dW <true,true>
            # where they are launched alone, GRP = gemm_f32_group_kernel (a block's dW slabs and its dX in one launch)
            lay = by_layout.setdefault(shape.split()[0], {"launches": 0, "total_us": 0.0})
            lay["launches"] += a["n"]
            lay["total_us"] += a["total_us"]
        elif a["unit"] == "byte":
            # a launch of ~10 us: what the event pair itself adds (KernelTimer.calibrate: measured around spin kernels
            # of known length) is a large share of the raw reading and is subtracted; the raw reading is kept beside it
            ent["avg_us_raw"] = round(a["raw_avg_us"], 2)
            ent["gbs"] = round(a["work"] / a["avg_us"] / 1e3, 1)
            ent["frac_of_peak"] = round(a["work"] / a["avg_us"] / 1e3 / HBM_PEAK_GBS, 4)
            ent["gbs_raw"] = round(a["work"] / a["raw_avg_us"] / 1e3, 1)
            ent["frac_of_peak_raw"] = round(a["work"] / a["raw_avg_us"] / 1e3 / HBM_PEAK_GBS, 4)
            gather = (a, ent)
        kernels.append(ent)
    for lay in by_layout.values():
        lay["avg_us"] = round(lay["total_us"] / max(lay["launches"], 1), 2)
        lay["total_us"] = round(lay["total_us"], 1)
    # HBM-side bytes per launch of a kernel family from the committed counter passes (profiles/pmc_traffic.json:
    # separate --pmc FETCH_SIZE / WRITE_SIZE runs of this benchmark's default configuration); null for any other
    # configuration, whose traffic those passes did not measure
    def pmc_traffic(prefix):
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        default_cfg = (args.model == "deepfm" and args.batch == 4096 and args.fields == 39 and args.dim == 16
                       and args.hidden == "1024,512,256" and not args.uniform)
        if not (os.path.exists(tp) and default_cfg):
            return None
        try:
            ks = [v for k, v in json.load(open(tp))["kernels"].items() if k.startswith(prefix)]     # prefix: str or tuple
            n = sum(v["launches"] for v in ks)
            return sum(v["launches"] * (v["read_bytes_per_launch"] + v["written_bytes_per_launch"]) for v in ks) / n if n else None
        except Exception:
            return None
    traffic = pmc_traffic(("gemm_f32_dma_kernel", "gemm_f32_group_kernel"))
    gather_traffic = pmc_traffic("gather_fused_kernel")
    roofline = None
    if gemm_us > 0:
        ach = gemm_flop / gemm_us / 1e6
        roofline = {"bound": "mfma", "kernel": "gemm_f32_dma_kernel / gemm_f32_group_kernel (v_mfma_f32_16x16x4_f32): every MLP GEMM "
                                               "launch of a step (NT = forward; GRP = a block's dW + dX in one launch)",
                    "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic,
                    "algorithmic_flop_per_step": gemm_flop / args.steps, "us_per_step": round(gemm_us / args.steps, 1),
                    "launches_per_step": sum(l["launches"] for l in by_layout.values()) / args.steps,
                    "avg_us_by_layout": by_layout, "measured_in": measured_in,
                    "event_pair_overhead_us_subtracted": round(timer.overhead_us, 2)}
    elif gather is not None:
        a, ent = gather
        roofline = {"bound": "hbm", "kernel": "gather_fused_kernel", "achieved": ent["gbs"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": ent["frac_of_peak"], "traffic": gather_traffic, "measured_in": measured_in}
    roofline_hbm = None
    if gather is not None:
        a, ent = gather
        roofline_hbm = {"bound": "hbm", "kernel": "gather_fused_kernel (K1)", "achieved": ent["gbs"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": ent["frac_of_peak"], "algorithmic_bytes_per_launch": a["work"],
                        "avg_us": ent["avg_us"], "traffic": gather_traffic,
                        "note": "duration = HIP-event pair around the launch minus what the pair itself adds "
                                f"({round(timer.overhead_us, 2)} us, measured around spin kernels of known length; raw "
                                f"reading {ent['avg_us_raw']} us); the rocprofv3 average of gather_fused_kernel in "
                                "profiles/*_bench_kernel_stats.csv is the same quantity (tools/collect_profiles.py "
                                "refuses a profile where the two differ by more than 15 %)"}

    # K3 (sparse backward + touched-rows update) against the HBM roofline: algorithmic bytes = per entry the index and
    # the gradient row (+ the embedding row re-read where the FM term is formed on the fly), per distinct row the
    # optimizer's reads and writes of w (and its moments), over the separately timed sort + reduce + update launches.
    roofline_hbm_scatter = None
    k3_us = sum(per_step.get(k, 0.0) for k in ("rs_sparse_bwd_sort", "rs_sparse_bwd_reduce", "rs_sparse_update",
                                               "rs_adam_catchup"))
    if not use_dist and fused_kind and args.mode == "sparse" and k3_us > 0 and hasattr(eng, "plan"):
        plan = eng.plan(args.batch)
        uniq = int(plan.counts()[:, 2].sum().item())             # distinct non-padding rows of the last batch
        d_emb = eng.d if eng.has_emb else 0
        per_entry = 8 + 4 * d_emb + (4 * d_emb if eng.has_fm else 0) + (4 if eng.has_lr else 0)
        n_state = {"SGD": 2, "Adagrad": 4}.get(args.optimizer, 6)
        per_row = n_state * 4 * (d_emb + (1 if eng.has_lr else 0))
        k3_bytes = args.batch * len(vocab) * per_entry + uniq * per_row
        roofline_hbm_scatter = {"bound": "hbm", "kernel": "K3: field_sort + chunk_reduce + segment_finalize + sparse_update",
                                "achieved": round(k3_bytes / k3_us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(k3_bytes / k3_us / 1e3 / HBM_PEAK_GBS, 4),
                                "algorithmic_bytes_per_step": k3_bytes, "distinct_rows": uniq, "us_per_step": round(k3_us, 1),
                                "note": "K3 = index sort + replay of skipped Adam steps (exact mode) + chunk reduce + "
                                        "segment finalize + touched-row update, each timed in a launch of its own"}

    cpu = None
    if rank == 0 and world == 1 and args.cpu_baseline:
        cpu = cpu_baseline(args, rec, hidden, idxs, labs)

    if rank == 0:
        total = args.batch * world * args.steps
        out = {
            "metric": "train samples/sec (node) DeepFM Criteo_x1 bs=4096" if args.model == "deepfm" and args.batch == 4096
                      else f"train samples/sec (node) {args.model} bs={args.batch}",
            "value": total / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.model} training step, Criteo_x1-shaped: "
                                   + ("39 fields (13 small-vocab + 26 log-spaced 10..1e6)" if args.fields == 39
                                      else "26 categorical fields (log-spaced 10..1e6)")
                                   + f", embedding_dim {args.dim}, MLP {hidden}, per-GPU batch {args.batch}",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "fields": len(vocab),
                       "embedding_dim": args.dim, "hidden": hidden, "optimizer": args.optimizer, "dropout": args.dropout,
                       "regular": args.regular,
                       "embedding_update": args.mode if args.mode == "dense" or args.optimizer not in ("Adam", "AdamW")
                       else ("sparse, exact: touched rows only, a row's skipped zero-gradient steps replayed before it is read "
                             "(the reference's dense optimizer)" if getattr(eng, "exact", False)
                             else "sparse, lazy: absent rows keep weights and moments"),
                       "index_distribution": "uniform" if args.uniform else "zipf(1.1)",
                       "distinct_batches": nb, "next_batch_announced": bool(use_next),
                       # exact Adam replays a returning row's skipped steps, so the step costs more once rows have been
                       # away for long (it levels off after some hundred steps): how old the tables are when timing starts
                       "steps_before_timed_region": args.warmup + aged,
                       "total_table_rows": sum(vocab), "hip_graph": bool(eng.use_graph),
                       "parallelism": ("single GPU" if not use_dist else
                                       "one rank of the N>1 step (--force-dist): "
                                       + ("replicated tables, RCCL all-gather" if args.tables == "replicated" and fused_kind
                                          else "row-sharded tables, every all-to-all / all-reduce through RCCL on a "
                                               "communicator of one")) if world == 1 else (
                           f"dp{world} + replicated tables (RCCL all-gather of (row, gradient) pairs)"
                           if args.tables == "replicated" and fused_kind
                           else f"dp{world} + row-sharded tables (RCCL all-to-all)")},
            "roofline": roofline, "roofline_hbm": roofline_hbm, "roofline_hbm_scatter": roofline_hbm_scatter,
            "cpu_baseline": cpu,
            "kernel_us_per_step": {k: round(v, 1) for k, v in per_step.items()},
            "kernels": kernels[:12], "final_loss": final_loss,
        }
        if exchange_check is not None:
            out["exchange_check"] = exchange_check
        if stdout_fd is not None:
            sys.stdout.flush()
            os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        if stdout_fd is not None:
            os.dup2(2, 1)                       # whatever the teardown prints is not part of the result
    if use_dist:
        import gc
        import torch.distributed as dist
        # graphs that captured collectives go before the communicator does
        eng.release_graphs() if hasattr(eng, "release_graphs") else getattr(eng, "_graphs", {}).clear()
        gc.collect()
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
