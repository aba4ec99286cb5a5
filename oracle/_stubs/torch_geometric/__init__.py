"""Stand-in for the `torch_geometric` package (see ../README.md). Test infrastructure only."""
__version__ = "0.0-mdno-stub"
