"""`torch_geometric.nn.conv.MessagePassing` stand-in (PyG >= 2.0 defaults).

propagate(edge_index, x=..., pseudo=...):
    x_j   = x.index_select(0, edge_index[0])              # flow = source_to_target
    msg   = self.message(x_j, pseudo)
    out   = scatter(msg, edge_index[1], dim=0, dim_size=x.size(0), reduce=self.aggr)
              "add":  index_add
              "mean": index_add / count.clamp(min=1)
              "max":  scatter amax (empty rows -> 0)
    return self.update(out, x)
"""
import torch


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2, **kwargs):
        super().__init__()
        assert flow == "source_to_target" and node_dim == -2
        self.aggr = aggr

    def propagate(self, edge_index, x, pseudo):
        n = x.size(0)
        src, dst = edge_index[0], edge_index[1]
        x_j = x.index_select(0, src)
        msg = self.message(x_j, pseudo)
        out = torch.zeros(n, msg.size(1), dtype=msg.dtype, device=msg.device)
        if self.aggr in ("add", "mean"):
            out.index_add_(0, dst, msg)
            if self.aggr == "mean":
                cnt = torch.zeros(n, dtype=msg.dtype, device=msg.device)
                cnt.index_add_(0, dst, torch.ones_like(dst, dtype=msg.dtype))
                out = out / cnt.clamp(min=1).unsqueeze(-1)
        elif self.aggr == "max":
            out = out.scatter_reduce(0, dst.unsqueeze(-1).expand_as(msg), msg,
                                     reduce="amax", include_self=False)
        else:
            raise ValueError(self.aggr)
        return self.update(out, x)
