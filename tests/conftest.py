"""pytest config: `gpu` marker + shared fixtures.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI symbol check, gloo world_size-2 tests.
`-m gpu`       : parity tests proper — every one calls the HIP kernels through the C-ABI
                 (libmdno.so) on cuda:0 and compares with the oracle / the golden vectors.
"""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

REPO = Path(__file__).resolve().parents[1]
GOLDEN = REPO / "tests" / "golden"
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def host_cores():
    """CPUs this process may really use: the affinity mask capped by the cgroup quota (a GPU box shows 256
    host threads in the mask and grants 16 CPUs; 256 torch threads on that quota crawl)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def load_golden(name):
    return np.load(GOLDEN / name, allow_pickle=True)


def golden_state_dict(z, prefix="p."):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def write_golden_trajectory(path, z):
    """The golden rollout fixture's trajectory as an .npz twin of the reference's HDF5 layout
    (flat contact maps + offsets: loads without pickle)."""
    from molecular_dynamics_neural_operator_amd.dataset import write_trajectory_npz
    frames = np.transpose(z["point_cloud"], (0, 2, 1))
    write_trajectory_npz(path, frames, list(z["contact_map"]), z["amino_acids"], z["rmsd"])
