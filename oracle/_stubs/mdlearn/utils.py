def log_latent_visualization(*a, **k):
    raise NotImplementedError("stub")
