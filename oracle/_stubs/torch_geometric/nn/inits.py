"""`torch_geometric.nn.inits` stand-in: `uniform` (bound 1/sqrt(size)) and recursive `reset`."""
import math


def uniform(size, tensor):
    if tensor is not None:
        bound = 1.0 / math.sqrt(size)
        tensor.data.uniform_(-bound, bound)


def reset(value):
    if hasattr(value, "reset_parameters"):
        value.reset_parameters()
    else:
        for child in value.children() if hasattr(value, "children") else []:
            reset(child)
