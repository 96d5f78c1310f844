"""Bare-name shim for QC/layers.py (`from layers import TransitionMLP, EdgeEncoderMLP, EdgeGraphConvolution`,
QC/layer_models.py:7, QC/models.py:7; the plain GCN layers of QC/layers.py:157-230 are the GCN ones)."""
from graph_odenet_amd.layers import FixedGraphConvolution, GraphConvolution  # noqa: F401
from graph_odenet_amd.qc_layers import EdgeGraphConvolution  # noqa: F401
from graph_odenet_amd.qc_models import MLP, EdgeEncoderMLP, MyLinear, NonLinear, TransitionMLP  # noqa: F401
