"""`imageio` stand-in: names only."""


def imread(*a, **k):
    raise NotImplementedError("stub")


def mimsave(*a, **k):
    raise NotImplementedError("stub")
