"""pytest configuration: registers the `gpu` marker, exposes the CPU oracles as fixtures, and -- for GPU sessions --
pins the native load order and leaves native evidence in the log if the process dies inside a C-ABI call.

The oracles (oracle/) are test infrastructure: `port` is our C++ restatement (always buildable with g++),
`ref` is the compiled unmodified reference (oracle/_ref/, prebuilt where /root/reference exists).

GPU sessions (any selected test carries the `gpu` marker), before the first test:
  1. a *preflight subprocess* creates, steps and destroys a 1-lane 1dx4f DCFR engine (the exact engine round 1's
     driver-side run died in).  A fresh process per attempt, native stderr captured; a transient failure of the box's
     first HIP process is retried (and reported), a persistent one fails the session with the native output in hand;
  2. torch, then librebel_hip.so are loaded before any oracle library (deterministic order, one HIP runtime);
  3. tests/native/libnative_bt.so installs a SIGABRT/SIGSEGV handler that prints the C backtrace before pytest's
     faulthandler prints the Python one.
pytest.ini runs with --capture=sys so that HIP / HSA / glibc diagnostics written to fd 2 reach the log.
"""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PREFLIGHT = r"""
import faulthandler, sys
faulthandler.enable()
import numpy as np
import torch
from rebel_amd import capi
print("preflight: torch", torch.__version__, "devices", capi.device_count(), flush=True)
p = capi.make_params(num_iters=8, max_depth=2, use_cfr=True, dcfr=True, dcfr_alpha=1.5, dcfr_beta=0.5, dcfr_gamma=2.0)
for _ in range(2):
    e = capi.Engine(1, 4, p, max_lanes=1)
    e.set_net_synthetic()
    e.reset([-1], [0], np.full((1, 2, 4), 0.25))
    e.multistep()
    assert np.isfinite(e.get(0, capi.GET_REGRETS)).all()
    e.close()
print("preflight: ok", flush=True)
"""


def _log(msg):
    sys.__stderr__.write(msg.rstrip("\n") + "\n")
    sys.__stderr__.flush()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(tryfirst=True)
def pytest_sessionfinish(session, exitstatus):
    """Native code (the compiled reference's progress prints) writes to C stdio; flush it before pytest prints its
    summary so that the `N passed` line stays the last line of the log."""
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def _gpu_preflight():
    """Runs the preflight in fresh processes; returns normally once one attempt succeeds."""
    attempts = []
    for attempt in range(3):
        env = dict(os.environ)
        if attempt:  # retries are verbose: HIP runtime trace of the failing call
            env["AMD_LOG_LEVEL"] = "3"
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, "-c", PREFLIGHT], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True, timeout=600)
            rc, out = r.returncode, r.stdout
        except subprocess.TimeoutExpired as ex:
            rc, out = -999, (ex.stdout or "") + "\n<preflight timed out>"
        attempts.append((rc, out))
        _log(f"[conftest] GPU preflight attempt {attempt + 1}: rc={rc} in {time.time() - t0:.1f}s")
        if rc == 0:
            if attempt:
                _log("[conftest] WARNING: an earlier preflight attempt failed on this box; its output:\n" +
                     attempts[0][1][-4000:])
            return
        _log(out[-6000:])
        time.sleep(5)
    pytest.fail("GPU preflight (1-lane 1dx4f DCFR engine in a fresh process) failed 3 times; last output:\n" +
                attempts[-1][1][-6000:], pytrace=False)


@pytest.fixture(scope="session", autouse=True)
def _gpu_session(request):
    if not any(item.get_closest_marker("gpu") for item in request.session.items):
        yield
        return
    _gpu_preflight()
    import ctypes

    import torch  # noqa: F401  (first: its bundled libamdhip64.so.7 serves the whole process)

    from rebel_amd import capi

    L = capi.lib()
    bt = os.path.join(ROOT, "tests", "native", "libnative_bt.so")
    if os.path.exists(bt):
        try:
            ctypes.CDLL(bt).native_bt_install(sys.__stderr__.fileno())
        except Exception as ex:  # diagnostics only
            _log(f"[conftest] native backtrace handler not installed: {ex}")
    hip_libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l})
    _log(f"[conftest] {L.rbl_build_info().decode()}; devices={capi.device_count()}; HIP/HSA runtimes mapped: {hip_libs}")
    yield


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
