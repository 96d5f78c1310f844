"""Shape buckets for QC mini-batches (BASELINE.json configs[3]).

Every QM9 mini-batch has its own node and edge counts (QC/datasets/utils.py:153-217 concatenates 20 molecules of 9-29
atoms), so a training step meets ~18 dense products whose row count it has never seen: the GPU libraries re-run their
kernel selection for each of them and the step costs twice what a repeated shape costs (3.5 vs 1.7 ms, DESIGN.md
section 5).  `pad_batch` rounds a collated batch up to the next bucket: it appends ONE extra graph made of isolated dummy
atoms (zero features) and dummy edges among them (zero features), so that node and edge counts are multiples of the
bucket sizes.  The padded batch is an ordinary batch - the reference's own model classes accept it unchanged - whose
first `n_graphs` output rows are those of the original batch: dummy atoms exchange messages only with each other,
readouts are per graph, and rows outside the loss carry no gradient.
"""
import torch


def pad_batch(x, edge_feat, Esrc, Etgt, batch, node_multiple=64, edge_multiple=128, n_graphs=None):
    """Returns (x, edge_feat, Esrc, Etgt, batch, n_graphs) with x.shape[0] a multiple of `node_multiple`, Esrc.numel()
    a multiple of `edge_multiple`, and the dummy graph numbered n_graphs (use `out[:n_graphs]`).  `Etgt` is the dense
    N x E matrix of the reference's collate, or the per-edge target index vector (int64[E]) a loader has before it
    builds that matrix - the same kind comes back.  `n_graphs`: the number of graphs of the batch when the caller knows it
    (one target row per graph); otherwise it is read from `batch` (one host synchronisation)."""
    n, e = x.shape[0], Esrc.numel()
    by_index = Etgt.dim() == 1 and not Etgt.is_floating_point()       # per-edge target index instead of the dense matrix
    if by_index:
        if Etgt.numel() != e:
            raise ValueError("pad_batch: one target index per edge")
    elif Etgt.dim() != 2 or tuple(Etgt.shape) != (n, e) or Etgt.layout != torch.strided:
        raise ValueError("pad_batch: Etgt must be the dense N x E incidence of the reference's collate (or the per-edge "
                         "target index vector)")
    if n_graphs is None:
        n_graphs = int(batch.max().item()) + 1 if n else 0
    e_pad = -(-max(e, 1) // edge_multiple) * edge_multiple
    n_pad = -(-(n + 1) // node_multiple) * node_multiple            # at least one dummy atom: dummy edges live on it
    dn, de = n_pad - n, e_pad - e
    dev = x.device
    x2 = torch.cat([x, x.new_zeros(dn, x.shape[1])])
    ef2 = torch.cat([edge_feat, edge_feat.new_zeros(de, edge_feat.shape[1])])
    src2 = torch.cat([Esrc, torch.full((de,), n, dtype=Esrc.dtype, device=dev)])
    if by_index:
        Etgt2 = torch.cat([Etgt, torch.full((de,), n, dtype=Etgt.dtype, device=dev)])
    else:
        Etgt2 = Etgt.new_zeros(n_pad, e_pad)
        Etgt2[:n, :e] = Etgt
        if de:
            Etgt2[n, e:] = 1.0
    batch2 = torch.cat([batch, torch.full((dn,), n_graphs, dtype=batch.dtype, device=dev)])
    return x2, ef2, src2, Etgt2, batch2, n_graphs


def prepare(Esrc, etgt, batch, n_nodes, n_graphs):
    """Loader-side preparation of a batch for the QC models: returns (edges, batch) to be passed in place of
    (Etgt, batch).  `etgt` is the per-edge target index (the loader has it before it builds the reference's dense
    N x E matrix); nothing here synchronises with the host, so a training loop that prepares its batches this way keeps
    the GPU queue full (the dense-matrix route costs a column arg-max over N x E, a validity check and a second
    synchronisation for the number of graphs - DESIGN.md section 5)."""
    from .qc_layers import prepared_edges
    batch._gode_n_graphs = int(n_graphs)
    return prepared_edges(Esrc, etgt, n_nodes), batch
