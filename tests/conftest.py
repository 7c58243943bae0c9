"""Test configuration.

* ``-m gpu``      : parity tests proper -- the hipcc-built libpcdm.so on cuda:0, checked against the oracle.
* ``-m "not gpu"``: oracle vs golden vectors / known answers, host logic, C-ABI export check, and the
  SAME kernel sources under the lane emulator (tests/emu) on tiny shapes.

Nothing here reads /root/reference at run time except tests explicitly skipped when it is absent.
"""
from __future__ import annotations

import sys
from dataclasses import dataclass
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """``-m "not gpu"`` (the CPU suite: ten of its tests each run a whole UNet under the single-threaded lane emulator, 30-70 s apiece)
    is spread over a few pytest-xdist workers unless the caller chose ``-n`` / ``-p no:xdist`` himself; the GPU suite is never
    parallelised (one GPU, timing-sensitive tests)."""
    import os
    if config.getoption("markexpr", "") != "not gpu" or os.environ.get("PYTEST_XDIST_WORKER") or os.environ.get("PCDM_TEST_SERIAL"):
        return None
    if not config.pluginmanager.hasplugin("xdist") or getattr(config.option, "numprocesses", None) is not None:
        return None
    from tests.emu import build_emu
    build_emu.build()                      # once, here: the workers must not race on the emulator build
    config.option.numprocesses = min(6, max(1, (os.cpu_count() or 2) // 2))
    config.option.dist = "load"
    return None


def pytest_configure(config):
    import os
    if os.environ.get("PYTEST_XDIST_WORKER"):   # share the host cores between the workers instead of oversubscribing them
        n = int(os.environ.get("PYTEST_XDIST_WORKER_COUNT", "1"))
        torch.set_num_threads(max(1, (os.cpu_count() or n) // n))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


@dataclass
class Backend:
    name: str
    device: torch.device

    @property
    def is_emu(self) -> bool:
        return self.name == "emu"

    def sync(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize()


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request) -> Backend:
    """Runs a kernel test twice: under the CPU lane emulator (tiny shapes) and on the GPU."""
    from pcdms_amd import _lib
    if request.param == "emu":
        from tests.emu import build_emu
        _lib.use_library(build_emu.load())
        return Backend("emu", torch.device("cpu"))
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _lib.load()
    assert not _lib.is_emulator()
    return Backend("gpu", torch.device("cuda:0"))


@pytest.fixture
def gpu_backend() -> Backend:
    from pcdms_amd import _lib
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _lib.load()
    assert not _lib.is_emulator()
    return Backend("gpu", torch.device("cuda:0"))
