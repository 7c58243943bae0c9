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


def pytest_configure(config):
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
