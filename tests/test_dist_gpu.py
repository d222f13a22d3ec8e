"""Row-sharded multi-process training against the single-GPU fused step on the union batch.

Two ranks share the one GPU of the test box and exchange through gloo with host staging (RCCL needs one
GPU per rank); every device kernel of the exchange is the real HIP one."""
import os

import numpy as np
import pytest
import torch

from dist_launch import run_ranks
from dist_worker import rank_batch

pytestmark = pytest.mark.gpu


def single_gpu_reference(kind, optimizer, world, generic=False):
    from dist_worker import GPU_BATCH, GPU_STEPS, GPU_VOCAB, build_gpu_model
    vocab, batch, steps = GPU_VOCAB, GPU_BATCH, GPU_STEPS
    m = build_gpu_model(kind, optimizer, batch * world)
    if generic:
        eng = m.use_sparse_step(max_norm=0.05, optimizer_type=optimizer)
    else:
        eng = m.use_fused_step(mode="sparse", max_norm=0.05, optimizer_type=optimizer)
    losses = []
    for s in range(steps):
        parts = [rank_batch(r + 10 * s, vocab, batch, zero_every=17) for r in range(world)]
        idx = torch.cat([p[0] for p in parts], 1).cuda()
        y = torch.cat([p[1] for p in parts], 0).cuda()
        losses.append(float(eng.step_tensors(idx, y).item()))
    return m, losses


def compare_weights(got, m, kind, optimizer):
    """Weights of the multi-rank run against the single-GPU run on the union batch: rtol 2e-4, atol 2e-6.  Towers with
    BatchNorm under Adam-type optimizers: the batch sum of a BatchNorm input's gradient is exactly zero, so the weight
    gradients of an input column that is (nearly) constant over the batch -- rank_batch gives field 1 a single key per
    rank -- are rounding noise, which Adam normalises to steps of up to lr (SGD runs of the same towers are held to the
    tight tolerance).  There, 85 % of every tensor must meet the tight tolerance and all of it half a step (lr = 1e-2)."""
    noisy = kind.endswith("_bn") and optimizer != "SGD"
    for k, v in m.state_dict().items():
        if not v.dtype.is_floating_point:
            continue
        a, b = torch.from_numpy(got[k]), v.cpu()
        if not noisy:
            torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-6, msg=lambda s, k=k: f"{k}: {s}")
            continue
        if k.endswith(".fc.bias"):
            continue                 # the bias in front of a BatchNorm: gradient exactly zero, all of it noise
        if k.endswith(".norm_layer.running_mean"):
            # ... and every step's batch mean carries that bias as it was then: bounded by the noise's size only
            assert float((a - b).abs().max()) <= 5e-3, f"{k}: greatest difference {float((a - b).abs().max())}"
            continue
        bad = (a - b).abs() > 2e-6 + 2e-4 * b.abs()
        assert float(bad.float().mean()) <= 0.15, f"{k}: {int(bad.sum())} of {bad.numel()} elements off"
        assert float((a - b).abs().max()) <= 5e-3, f"{k}: greatest difference {float((a - b).abs().max())}"


@pytest.mark.parametrize("kind,optimizer,extra", [("deepfm", "SGD", ""), ("deepfm", "Adam", ""), ("fm", "Adagrad", ""),
                                                  ("lr", "AdamW", ""), ("dnn", "Adam", ""),
                                                  ("deepfm", "Adam", "onestream"), ("fm", "SGD", "onestream"),
                                                  ("deepfm_bn", "SGD", ""), ("deepfm_bn", "Adam", ""),
                                                  ("dnn_bn", "SGD", "onestream"),
                                                  ("dcn", "Adam", ""), ("dcn_bn", "SGD", "onestream"),
                                                  ("pnn", "Adam", ""), ("pnn_bn", "SGD", "")])
def test_two_ranks_match_single_gpu_union_batch(kind, optimizer, extra, tmp_path):
    """Default: the dense all-reduce on a second stream and communicator beside the gradient exchange; onestream:
    everything on one stream (the dense gradient's norm partials then ride in the owner's reduction).  *_bn: towers with
    BatchNorm in every block (the reference's default): batch statistics over the union of the ranks' batches, running
    statistics included in the comparison.  dcn / pnn: the hand-scheduled DCN and PNN-inner steps (cross network
    parameters and the padded first-layer weight ride in the flat buffer's all-reduce)."""
    world = 2
    codes, outs = run_ranks(world, ["gpu", str(tmp_path), kind, optimizer] + ([extra] if extra else []), timeout=200)
    assert codes == [0] * world, "\n".join(outs)
    got = np.load(tmp_path / "state.npz")
    m, losses = single_gpu_reference(kind, optimizer, world)
    # mean loss over the union batch = mean of the ranks' local means (equal batch sizes)
    lr = np.mean([np.load(tmp_path / f"loss{r}.npy") for r in range(world)], 0)
    np.testing.assert_allclose(lr, np.array(losses), rtol=2e-5)
    compare_weights(got, m, kind, optimizer)


@pytest.mark.parametrize("kind,optimizer", [("pnn", "Adam"), ("dcn", "SGD"), ("nfm", "Adagrad"), ("afm", "AdamW"),
                                            ("ffm", "Adam"), ("ffm", "SGD"), ("dcn_bn", "SGD"), ("pnn_bn", "SGD")])
def test_two_ranks_generic_step_match_single_gpu_union_batch(kind, optimizer, tmp_path):
    """DistributedSparseTrainer (model forward under autograd, sharded lookups) against SparseTrainer on the union."""
    world = 2
    codes, outs = run_ranks(world, ["gpu", str(tmp_path), kind, optimizer, "generic"], timeout=200)
    assert codes == [0] * world, "\n".join(outs)
    got = np.load(tmp_path / "state.npz")
    m, losses = single_gpu_reference(kind, optimizer, world, generic=True)
    lr = np.mean([np.load(tmp_path / f"loss{r}.npy") for r in range(world)], 0)
    np.testing.assert_allclose(lr, np.array(losses), rtol=2e-5)
    compare_weights(got, m, kind, optimizer)


@pytest.mark.parametrize("kind,optimizer,extra", [("deepfm", "Adam", "replicated"), ("fm", "SGD", "replicated"),
                                                  ("lr", "AdamW", "replicated"), ("deepfm", "Adagrad", "replicated"),
                                                  ("deepfm_bn", "SGD", "replicated")])
def test_two_ranks_replicated_tables_match_single_gpu_union_batch(kind, optimizer, extra, tmp_path):
    """ReplicatedTablesTrainer: every rank keeps whole tables and merges the all-gathered (row, gradient) pairs."""
    world = 2
    codes, outs = run_ranks(world, ["gpu", str(tmp_path), kind, optimizer, extra], timeout=200)
    assert codes == [0] * world, "\n".join(outs)
    got = np.load(tmp_path / "state.npz")
    m, losses = single_gpu_reference(kind, optimizer, world)
    lr = np.mean([np.load(tmp_path / f"loss{r}.npy") for r in range(world)], 0)
    np.testing.assert_allclose(lr, np.array(losses), rtol=2e-5)
    compare_weights(got, m, kind, optimizer)


@pytest.mark.parametrize("kind,optimizer", [("deepfm", "Adam"), ("deepfm", "SGD"), ("fm", "AdamW"), ("lr", "Adagrad"),
                                            ("dnn", "Adam"), ("deepfm_bn", "SGD"), ("dcn", "Adam"), ("pnn", "SGD")])
def test_one_rank_rccl_graph_matches_eager(kind, optimizer, tmp_path):
    """The real RCCL calls on a one-rank group: the step issued eagerly, the whole step as ONE HIP graph with the
    collectives captured inside, that graph with the dense all-reduce on the second stream and communicator, and (eagerly
    and from graphs) with every next batch announced, whose key exchange then runs on that second stream -- the same
    losses and weights bit for bit -- and all of them against the single-GPU fused step."""
    codes, outs = run_ranks(1, ["gpu1", str(tmp_path), kind, optimizer], timeout=200)
    if codes != [0] and any("hipErrorCapturedEvent" in o and "watchdog" in o for o in outs):
        # torch's process-group watchdog thread queried an event that a capture holds and aborted the process: a race
        # inside torch.distributed between its watchdog and any stream capture (seen once in several hundred captures,
        # with the six trainers this worker builds one after the other), not a result of this package's kernels
        pytest.xfail("torch.distributed watchdog aborted the worker on a captured event")
    assert codes == [0], "\n".join(outs)
    ref = np.load(tmp_path / "state_eager.npz")
    for name in ("graph", "graph_overlap", "eager_ahead", "graph_ahead", "replicated_graph"):
        got = np.load(tmp_path / f"state_{name}.npz")
        assert sorted(got.files) == sorted(ref.files)
        for k in ref.files:
            if name.startswith("replicated"):      # another way to the same sums
                np.testing.assert_allclose(got[k], ref[k], rtol=2e-4, atol=2e-6, err_msg=f"{name}: {k}")
            else:
                assert np.array_equal(got[k], ref[k]), f"{name}: {k}"
    m, losses = single_gpu_reference(kind, optimizer, 1)
    np.testing.assert_allclose(ref["__losses__"], np.array(losses), rtol=2e-5)
    for k, v in m.state_dict().items():
        if v.dtype.is_floating_point:
            torch.testing.assert_close(torch.from_numpy(ref[k]), v.cpu(), rtol=2e-4, atol=2e-6, msg=lambda s, k=k: f"{k}: {s}")


@pytest.mark.parametrize("extra", [["--force-dist", "1"], ["--gpus", "2", "--rehearse-on-one-gpu", "1"]])
def test_bench_stdout_is_one_json_line_when_a_communicator_comes_up(extra):
    """RCCL prints a version banner and gloo its connection lines on the C-level stdout: the benchmark must keep them out
    of its own stdout, which carries exactly the one result line (the real RCCL path with one rank; the two-rank
    rehearsal through the self-launcher)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "512",
                        "--hidden", "64,32", "--cpu-baseline", "0", "--event-pass", "0", "--distinct-batches", "2", *extra],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    assert d["exchange_check"] == "ok" and d["steps"] == 2
    assert d["n_gpus"] == (2 if "--gpus" in extra else 1)
