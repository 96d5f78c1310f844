"""The bare-name shims make `import layers / models / torchdiffeq` resolve to this package, with the
reference's constructor signatures and state_dict keys (enumerated in SURVEY.md §8b).  CPU only: no
kernel is launched (construction + introspection)."""
import importlib
import inspect
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "graph_odenet_amd", "dropin")


@pytest.fixture
def shim_path():
    saved_path, saved_mods = list(sys.path), {k: sys.modules.get(k) for k in ("layers", "models", "mpnn", "torchdiffeq")}
    for k in saved_mods:
        sys.modules.pop(k, None)

    def use(variant):
        for k in saved_mods:
            sys.modules.pop(k, None)
        sys.path[:] = [os.path.join(DROPIN, variant), DROPIN] + saved_path
    yield use
    sys.path[:] = saved_path
    for k, v in saved_mods.items():
        sys.modules.pop(k, None)
        if v is not None:
            sys.modules[k] = v


def test_gcn_shims(shim_path):
    shim_path("GCN")
    models = importlib.import_module("models")
    layers = importlib.import_module("layers")
    td = importlib.import_module("torchdiffeq")
    assert models.__file__.startswith(DROPIN) and layers.__file__.startswith(DROPIN)
    assert list(inspect.signature(td.odeint_adjoint).parameters)[:5] == ["func", "y0", "t", "rtol", "atol"]
    m = models.ODEGCN3(nfeat=10, nhid=16, nclass=3, dropout=0.5)          # GCN/train_res.py:122-125 kwargs
    assert list(m.state_dict().keys()) == [
        "gc1.weight", "gc1.bias", "gc2.odefunc.norm1.weight", "gc2.odefunc.norm1.bias",
        "gc2.odefunc.gc1.weight", "gc2.odefunc.gc1.bias", "gc3.weight", "gc3.bias"]
    assert m.gc2.odefunc.gc1.weight.shape == (17, 16) and m.gc2.tol == 1e-5
    m.nfe = 0                                                              # GCN/train_res.py:64
    assert m.nfe == 0 and m.gc2.odefunc.nfe == 0
    assert repr(layers.GraphConvolution(4, 2)) == "GraphConvolution (4 -> 2)"
    assert layers.GraphConvolution(4, 2, bias=False).bias is None
    # U(-1/sqrt(out), 1/sqrt(out)) init (GCN/layers.py:25-29)
    w = layers.GraphConvolution(300, 16).weight
    assert w.abs().max() <= 0.25 and w.abs().max() > 0.2
    for name in ("GCN", "GCN3", "RGCN3", "RGCN3norm", "RGCN3fullnorm", "ODEGCN3fullnorm"):
        getattr(models, name)(nfeat=10, nhid=16, nclass=3, dropout=0.5)


def test_gat_shims(shim_path):
    shim_path("GAT")
    models = importlib.import_module("models")
    layers = importlib.import_module("layers")
    lay = layers.GraphConvolution(5, 4)
    assert lay.f.weight.shape == (4, 10) and lay.w.weight.shape == (1, 10) and lay.eps == 1e-6
    assert list(inspect.signature(lay.forward).parameters) == ["x", "src", "tgt", "Mtgt"]
    m = models.ODEGCN3(nfeat=10, nhid=16, nclass=3, dropout=0.5)
    assert "gc2.odefunc.gc1.f.weight" in m.state_dict() and "gc2.odefunc.gc1.w.bias" in m.state_dict()
    assert list(inspect.signature(m.forward).parameters) == ["x", "src", "tgt", "Mtgt"]
    assert list(inspect.signature(m.gc2.odefunc.set_adj).parameters) == ["src", "tgt", "Mtgt"]


def test_qc_shims(shim_path):
    shim_path("QC")
    mpnn = importlib.import_module("mpnn")
    layers = importlib.import_module("layers")
    m = mpnn.MPNN_enn_edge(5, 73)
    assert m.T == 8 and isinstance(m.update_net, torch.nn.GRUCell) and m.update_net.input_size == 146
    m.set_T(3)
    assert m.T == 3
    assert list(inspect.signature(m.forward).parameters) == ["x", "Esrc", "Etgt", "edge_data"]
    e = layers.EdgeGraphConvolution(13, 73, node_layers=1, edge_layers=1, bias=True)
    assert e.weight.shape == (13, 73) and list(inspect.signature(e.forward).parameters) == ["input", "Esrc", "Etgt", "edge_data"]


REFERENCE_ZOO = ["GCN", "RGCN2", "ODEGCN2", "GCN3", "GCN3norm", "RGCN3", "RGCN3norm", "RGCN3fullnorm", "ODEfunc", "ODEBlock",
                 "ODEGCN3", "ODEGCN3fullnorm", "GCNK", "GCNKnorm", "RESK1", "RESK2", "RESK", "RESK1norm", "RESK2norm",
                 "RESKnorm", "ODEK1", "ODEfunc2", "ODEK2"]      # `grep ^class */models.py` of the reference, every variant


@pytest.mark.parametrize("variant,first_keys", [
    ("GCN", ["gc1.weight", "gc1.bias"]), ("GCN-sum", ["gc1.weight", "gc1.bias"]),
    ("GCN-dense-paper", ["gc1.weight", "gc1.bias"]),
    ("GCN-mlp-sum", ["gc1.mlp.layers.0.linear.weight", "gc1.mlp.layers.0.linear.bias"]),
    ("GAT", ["gc1.f.weight", "gc1.f.bias"])])
def test_every_variant_exports_the_reference_zoo(shim_path, variant, first_keys):
    """`import models` / `import layers` from each variant's shim directory: every class name the reference's
    models.py defines is there (train_res.py / train_layers.py index `models.<Name>` at import time), assembled from
    that variant's layers."""
    shim_path(variant)
    models = importlib.import_module("models")
    layers = importlib.import_module("layers")
    for name in REFERENCE_ZOO:
        assert hasattr(models, name), "%s lacks %s" % (variant, name)
    assert hasattr(layers, "GraphConvolution") and hasattr(layers, "FixedGraphConvolution")
    m = models.ODEGCN3(nfeat=7, nhid=8, nclass=3, dropout=0.5)
    assert list(m.state_dict().keys())[:2] == first_keys
    assert isinstance(m.gc1, layers.GraphConvolution)
    deep = models.RESK2norm(nfeat=7, nhid=8, nclass=3, dropout=0.5, nlayers=5)
    assert len(deep.gcs) == 5 and len(deep.norms) == 3 and isinstance(deep.gcs[2], layers.GraphConvolution)
    if variant == "GCN-mlp-sum":
        assert all(hasattr(layers, n) for n in ("MyLinear", "NonLinear", "MLP"))
    expected = ["x", "src", "tgt", "Mtgt"] if variant == "GAT" else ["x", "adj"]
    assert list(inspect.signature(m.forward).parameters) == expected


def test_qc_shims_cover_the_reference_imports(shim_path):
    """Every bare-name import of QC/train_egcn.py, train_egcn_multitask.py, layer_models.py and models.py resolves."""
    shim_path("QC")
    for k in ("layer_models", "set2set", "torch_scatter"):
        sys.modules.pop(k, None)
    lm = importlib.import_module("layer_models")
    for name in ("EdgeGCN_K_Sum", "EdgeGCN_K_Set2Set", "MPNN_ENN_K_Sum", "MPNN_ENN_K_Set2Set", "EdgeRES1_K_Set2Set",
                 "UnimplementedModel", "RESKnorm", "get_output_function"):      # QC/train_egcn.py:85-94
        assert hasattr(lm, name), name
    with pytest.raises(NotImplementedError):
        lm.UnimplementedModel()
    layers = importlib.import_module("layers")
    for name in ("MyLinear", "NonLinear", "MLP", "TransitionMLP", "EdgeEncoderMLP", "EdgeGraphConvolution",
                 "GraphConvolution", "FixedGraphConvolution"):                  # classes of QC/layers.py in use
        assert hasattr(layers, name), name
    models = importlib.import_module("models")
    for name in ("MPNN_ENN_Sum", "MPNN_ENN_Set2Set", "EdgeGCN3_Sum", "EdgeGCN3_Set2Set"):   # QC/models.py
        assert hasattr(models, name), name
    assert hasattr(importlib.import_module("mpnn"), "MPNN_enn_edge")
    assert hasattr(importlib.import_module("set2set"), "Set2Set")
    assert hasattr(importlib.import_module("torch_scatter"), "scatter_add")
    for k in ("layer_models", "set2set", "torch_scatter"):
        sys.modules.pop(k, None)
    m = models.EdgeGCN3_Set2Set(13, 5, 8, 4, processing_steps=2)
    assert "ee3.mlp.layers.0.linear.weight" in m.state_dict() or any(k.startswith("ee3.") for k in m.state_dict())
