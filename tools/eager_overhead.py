import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cProfile, pstats
from recommendsystem_amd import data, models, ops
rec = data.criteo_x1_recorder(16, 4096)
m = models.DeepFM(rec, 1e-3, 0, hidden_dim_list=[1024,512,256], normalization=None)
m.train()
vocab=[rec.feature_map[c]["vocab_size"] for c in rec.features]
idxs, labs = data.synthetic_batches(vocab, 4096, 4, "cuda:0")
for overlap in (True, False):
    eng = m.use_fused_step(mode="sparse", use_graph=False, overlap=overlap)
    for i in range(5): eng.step_tensors(idxs[i%4], labs[i%4])
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(30): eng.step_tensors(idxs[i%4], labs[i%4])
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"overlap={overlap}: host {1e3*(t1-t0)/30:.3f} ms/step, total {1e3*(t2-t0)/30:.3f} ms/step")
pr=cProfile.Profile(); pr.enable()
for i in range(20): eng.step_tensors(idxs[i%4], labs[i%4])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
