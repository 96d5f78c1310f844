"""The GCN-dense-paper variant of the reference (GCN-dense-paper/layers.py, models.py): the same layers and model
zoo as GCN/ with (a) Glorot-uniform weights (gain for relu) and zero biases (layers.py:26-29,63-66), (b) dropout on
the input features at the top of every model (models.py:17 and the like) and (c) a dense, symmetrically normalised
adjacency from its utils.py - which graph.as_graph accepts like any other format.  Compute is the GCN path."""
import torch
import torch.nn as nn

from . import layers, models


def _glorot_relu(layer):
    torch.nn.init.xavier_uniform_(layer.weight, gain=nn.init.calculate_gain('relu'))
    if layer.bias is not None:
        torch.nn.init.constant_(layer.bias, 0)


class GraphConvolution(layers.GraphConvolution):
    def reset_parameters(self):
        _glorot_relu(self)


class FixedGraphConvolution(layers.FixedGraphConvolution):
    def reset_parameters(self):
        _glorot_relu(self)


class ODEfunc(models.ODEfunc):
    layer_cls = FixedGraphConvolution


class ODEfunc2(models.ODEfunc2):
    layer_cls = FixedGraphConvolution


ODEBlock = models.ODEBlock


class DensePaperKit:
    GraphConvolution = GraphConvolution
    ODEfunc = ODEfunc
    ODEfunc2 = ODEfunc2
    input_dropout = True


models.rebind_zoo(globals(), __name__, DensePaperKit, what="GCN-dense-paper")
