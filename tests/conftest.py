import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """The C oracle and the CPU emulation harness are test infrastructure: build them if missing
    (gcc/g++ only).  The HIP library is built by __graft_entry__.build(); tests never rebuild it on the
    GPU box (it travels prebuilt)."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "libp2e_oracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libp2e_oracle.so"])
    if not os.path.exists(os.path.join(ROOT, "tests", "emu", "libp2e_emu.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    if not os.path.exists(os.path.join(ROOT, "plonky2-ecdsa_amd", "libp2e_hip.so")):
        import __graft_entry__ as ge
        ge.build()
    yield
