import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Every rendezvous in the suite is on 127.0.0.1: pin gloo / RCCL bootstrap to the loopback
# interface so that neither tries to resolve the container hostname (which may not resolve and
# then stalls for minutes before falling back).  Inherited by the spawned rank processes.
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle's oneDNN convolutions are fastest at ~16 threads and an order of magnitude slower
    # at the 128+ threads torch picks by default on the GPU box's 256 logical CPUs (tools/cpu_probe.py)
    import torch
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
