"""How much replay of skipped Adam steps a batch asks for as training goes on: distinct rows, rows that are behind, rows
whose moments are not zero (their replay is a loop), their gaps, and the time of the stand-alone catch-up launch.

    python tools/replay_load.py"""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from recommendsystem_amd import data, models, ops
from recommendsystem_amd.base_model import random_all
random_all(2021)
vs = data.criteo_x1_vocab_sizes()
rec = data.criteo_x1_recorder(embedding_dim=16, batch_size=4096, vocab_sizes=vs)
m = models.DeepFM(rec, 1e-3, 0, hidden_dim_list=[1024, 512, 256], normalization=None, optimizer_type="Adam")
m.train()
eng = m.use_fused_step(mode="sparse", use_graph=False)
N = 330
idxs, labs = data.synthetic_batches(vs, 4096, N, "cuda", seed=2021)
te = eng._table_entry(eng.tab)
rs = te["rs"]
for s in range(N):
    if s in (20, 60, 150, 300, 320):
        idx = idxs[s]
        rows = (eng.layout.row_offset_t.view(-1, 1) + idx).view(-1)
        rows = torch.unique(rows[idx.view(-1) > 0])
        last = rs.last[rows].long()
        gap = int(eng.step_count.item()) - last
        mv = (te["m"][rows] != 0).any(1)
        ret = (gap > 0) & mv
        g = gap[ret].float()
        print(f"step {s}: distinct {rows.numel()}, stale {(gap > 0).sum().item()}, returning (m != 0) {ret.sum().item()}, "
              f"mean gap {g.mean().item():.1f}, max {g.max().item():.0f}, sum min(gap,185) {g.clamp(max=185).sum().item():.0f}", flush=True)
        # time the stand-alone catch-up on a copy of the state
        snap = [(t, t.clone()) for t in eng._mutable_state()]
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with ops.use_workspace(eng.ws):
            a.record(); eng._embed_catch_up(idx, 0); b.record()
        torch.cuda.synchronize()
        print(f"   rs_adam_catchup: {a.elapsed_time(b) * 1e3:.1f} us", flush=True)
        for t, c in snap:
            t.copy_(c)
    eng.step_tensors(idxs[s], labs[s])
