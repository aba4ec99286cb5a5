"""`torch_geometric.data` stand-in: attribute-bag `Data` + a do-nothing `DataLoader` name."""
import torch


class Data:
    """Attribute container with the handful of `Data` behaviours the reference touches."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def _tensor_items(self):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                yield k, v

    def to(self, device, *args, **kwargs):
        for k, v in self._tensor_items():
            setattr(self, k, v.to(device, *args, **kwargs))
        return self

    def __inc__(self, key, value, *args, **kwargs):
        return 0

    def __repr__(self):
        body = ", ".join(f"{k}={list(v.shape)}" for k, v in self._tensor_items())
        return f"{type(self).__name__}({body})"


class DataLoader:  # name only; the golden generator never iterates one
    def __init__(self, *a, **k):
        raise NotImplementedError("stub")
