"""qc_batch.pad_batch on the CPU: the padded batch is an ordinary batch for the REFERENCE formulas (oracle restatement of
QC/mpnn.py:19-32 and QC/layers.py:136-149) whose first n_graphs outputs are those of the original batch."""
import torch

from graph_odenet_amd.qc_batch import pad_batch
from graph_odenet_amd.synth import qm9_like_batch
from oracle import layers_ref as R


def test_pad_batch_shapes_and_isolation():
    x, ef, Esrc, Etgt, batch = qm9_like_batch(5, seed=2)
    n, e = x.shape[0], Esrc.numel()
    xp, efp, srcp, Etp, bp, nb = pad_batch(x, ef, Esrc, Etgt, batch, node_multiple=32, edge_multiple=64)
    assert nb == 5 and xp.shape[0] % 32 == 0 and srcp.numel() % 64 == 0 and xp.shape[0] > n
    assert Etp.shape == (xp.shape[0], srcp.numel()) and efp.shape[0] == srcp.numel() and bp.shape[0] == xp.shape[0]
    assert torch.equal(xp[:n], x) and torch.equal(srcp[:e], Esrc) and torch.equal(Etp[:n, :e], Etgt) and torch.equal(bp[:n], batch)
    assert bool((xp[n:] == 0).all()) and bool((efp[e:] == 0).all()) and bool((bp[n:] == nb).all())
    # dummy edges run from a dummy atom to a dummy atom; no real atom receives or sends along them
    assert bool((srcp[e:] >= n).all()) and bool((Etp[:n, e:] == 0).all()) and bool((Etp[n:, :e] == 0).all())
    assert bool((Etp.sum(0) == 1).all())
    # already a multiple: still one dummy atom, no dummy edge needed
    k = Esrc.numel()
    _, _, s2, E2, _, _ = pad_batch(x, ef, Esrc, Etgt, batch, node_multiple=1, edge_multiple=k)
    assert s2.numel() == k and E2.shape == (n + 1, k)


def test_padded_batch_gives_the_reference_formulas_the_same_outputs():
    torch.manual_seed(0)
    h = 12
    x, ef, Esrc, Etgt, batch = qm9_like_batch(4, seed=7)
    n, e = x.shape[0], Esrc.numel()
    xp, efp, srcp, Etp, bp, nb = pad_batch(x, ef, Esrc, Etgt, batch, node_multiple=32, edge_multiple=64)
    W_in = torch.randn(13, h) * 0.3
    enc = torch.nn.Sequential(torch.nn.Linear(5, 20), torch.nn.ReLU(), torch.nn.Linear(20, h * h))
    gru = torch.nn.GRUCell(2 * h, h)
    gcw, gcb = torch.randn(h, h) * 0.2, torch.randn(h) * 0.1
    with torch.no_grad():
        def run(x_, ef_, src_, Et_):
            A = enc(ef_).reshape(-1, h, h)
            y = R.mpnn_enn_edge(x_ @ W_in, src_, Et_, A, gru, 2)
            return y, R.edge_graph_convolution(y, src_, Et_, A, gcw, gcb)
        y0, z0 = run(x, ef, Esrc, Etgt)
        y1, z1 = run(xp, efp, srcp, Etp)
    assert (y1[:n] - y0).abs().max() < 1e-6 and (z1[:n] - z0).abs().max() < 1e-6
    # per-graph sum readout: the dummy graph is row nb, the real graphs are unchanged
    s0 = torch.zeros(4, h).index_add_(0, batch, z0)
    s1 = torch.zeros(nb + 1, h).index_add_(0, bp, z1)
    assert (s1[:nb] - s0).abs().max() < 1e-5


def test_pad_batch_accepts_the_target_index_vector():
    x, ef, Esrc, Etgt, batch = qm9_like_batch(3, seed=5)
    etgt = Etgt.argmax(0)
    a = pad_batch(x, ef, Esrc, Etgt, batch, 32, 64)
    b = pad_batch(x, ef, Esrc, etgt, batch, 32, 64)
    assert b[3].dim() == 1 and b[3].numel() == a[3].shape[1]
    assert torch.equal(a[3].argmax(0), b[3]) and torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[4], b[4])
