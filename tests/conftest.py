"""pytest configuration: registers the `gpu` marker and exposes the CPU oracles as fixtures.

The oracles (oracle/) are test infrastructure: `port` is our C++ restatement (always buildable with g++),
`ref` is the compiled unmodified reference (oracle/_ref/, prebuilt where /root/reference exists).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def port():
    from oracle import orc

    return orc.Oracle("port")


@pytest.fixture(scope="session")
def ref():
    from oracle import orc

    if not orc.have_ref():
        pytest.skip("oracle/_ref/libref_driver.so not built (needs /root/reference: `make -C oracle ref`)")
    return orc.Oracle("ref")


@pytest.fixture(scope="session", params=["port", "ref"])
def any_oracle(request):
    from oracle import orc

    if request.param == "ref" and not orc.have_ref():
        pytest.skip("compiled reference not available")
    return orc.Oracle(request.param)
