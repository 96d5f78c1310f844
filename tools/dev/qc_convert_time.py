import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd.synth import qm9_like_batch
from graph_odenet_amd import qc_layers, qc_models
dev = torch.device("cuda:0")
batches = [qm9_like_batch(20, seed=s, device=dev) for s in range(40)]
torch.cuda.synchronize()
for name, fn in (("_EdgeSet(Esrc, Etgt)", lambda b: qc_layers._EdgeSet(b[2], b[3])), ("_Segments(batch)", lambda b: qc_models._Segments(b[4]))):
    fn(batches[0]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches[1:]:
        fn(b)
    torch.cuda.synchronize()
    print("%-22s %.3f ms per fresh batch" % (name, (time.perf_counter() - t0) / 39 * 1e3))
