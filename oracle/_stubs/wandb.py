"""`wandb` stand-in: names only (the golden generator never logs)."""
run = None
config = None


def init(*a, **k):
    raise NotImplementedError("stub")


def log(*a, **k):
    raise NotImplementedError("stub")
