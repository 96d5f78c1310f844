"""GAT-family ODE models (reference: GAT/models.py - the GCN zoo with (x, src, tgt, Mtgt) plumbing)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .gat_layers import FixedGraphConvolution, GraphConvolution
from .odeint import odeint_adjoint as odeint


class ODEfunc(nn.Module):
    """GAT/models.py:161-179."""

    def __init__(self, dim):
        super(ODEfunc, self).__init__()
        self.norm1 = nn.GroupNorm(min(32, dim), dim)
        self.gc1 = FixedGraphConvolution(dim + 1, dim)
        self.nfe = 0

    def set_adj(self, src, tgt, Mtgt):
        self.gc1.set_adj(src, tgt, Mtgt)

    def forward(self, t, x):
        self.nfe += 1
        x = self.norm1(x)
        tt = torch.ones_like(x[:, :1]) * t
        ttx = torch.cat([tt, x], 1)
        return F.relu(self.gc1(ttx))


class ODEBlock(nn.Module):
    """GAT/models.py:181-201."""

    def __init__(self, odefunc, tol=1e-5, method=None, step_size=None):
        super(ODEBlock, self).__init__()
        self.odefunc = odefunc
        self.integration_time = torch.tensor([0, 1]).float()
        self.tol = tol
        self.method = method
        self.step_size = step_size

    def forward(self, x, src, tgt, Mtgt):
        self.integration_time = self.integration_time.type_as(x)
        self.odefunc.set_adj(src, tgt, Mtgt)
        options = {"step_size": self.step_size} if self.step_size is not None else None
        out = odeint(self.odefunc, x, self.integration_time, rtol=self.tol, atol=self.tol,
                     method=self.method, options=options)
        return out[1]

    @property
    def nfe(self):
        return self.odefunc.nfe

    @nfe.setter
    def nfe(self, value):
        self.odefunc.nfe = value


class GCN3(nn.Module):
    """GAT/models.py:66-81."""

    def __init__(self, nfeat, nhid, nclass, dropout):
        super(GCN3, self).__init__()
        self.gc1 = GraphConvolution(nfeat, nhid)
        self.gc2 = GraphConvolution(nhid, nhid)
        self.gc3 = GraphConvolution(nhid, nclass)
        self.dropout = dropout

    def forward(self, x, src, tgt, Mtgt):
        x = F.relu(self.gc1(x, src, tgt, Mtgt))
        x = F.dropout(x, self.dropout, training=self.training)
        x = F.relu(self.gc2(x, src, tgt, Mtgt))
        x = F.dropout(x, self.dropout, training=self.training)
        x = self.gc3(x, src, tgt, Mtgt)
        return F.log_softmax(x, dim=1)


class ODEGCN3(nn.Module):
    """GAT/models.py:204-226."""

    def __init__(self, nfeat, nhid, nclass, dropout, method=None, step_size=None, tol=1e-5):
        super(ODEGCN3, self).__init__()
        self.gc1 = GraphConvolution(nfeat, nhid)
        self.gc2 = ODEBlock(ODEfunc(nhid), tol=tol, method=method, step_size=step_size)
        self.gc3 = GraphConvolution(nhid, nclass)
        self.dropout = dropout

    def forward(self, x, src, tgt, Mtgt):
        x = F.relu(self.gc1(x, src, tgt, Mtgt))
        x = F.dropout(x, self.dropout, training=self.training)
        x = self.gc2(x, src, tgt, Mtgt)
        x = self.gc3(x, src, tgt, Mtgt)
        return F.log_softmax(x, dim=1)

    @property
    def nfe(self):
        return self.gc2.nfe

    @nfe.setter
    def nfe(self, value):
        self.gc2.nfe = value
