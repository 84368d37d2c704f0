"""GPU parity tests (P2): the fused MFMA value-net forward against (a) the committed torch-CPU outputs of the
reference's own Net2 class on reference-produced queries (tests/golden/net2_1d6f.npz) and (b) a float64 numpy
restatement of Net2 (cfvpy/models.py:64-94) with O(1) outputs.  Tolerance: 1e-5 absolute (north_star)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = 1e-5


def _np_net(q, layers, ln, w_out, b_out, eps=1e-5):
    from scipy.special import erf

    x = q.astype(np.float64)
    for i, (w, b) in enumerate(layers):
        x = x @ w.astype(np.float64).T + b
        if ln is not None:
            g, o = ln[i]
            mu = x.mean(-1, keepdims=True)
            var = ((x - mu) ** 2).mean(-1, keepdims=True)
            x = (x - mu) / np.sqrt(var + eps) * g + o
        x = 0.5 * x * (1 + erf(x / np.sqrt(2)))
    return x @ w_out.astype(np.float64).T + b_out


def _golden():
    import os

    return np.load(os.path.join(os.path.dirname(__file__), "golden", "net2_1d6f.npz"))


def _engine(d, f):
    from rebel_amd import capi

    return capi.Engine(d, f, capi.make_params(num_iters=4, use_cfr=True))


def test_net2_vs_torch_cpu_golden():
    g = _golden()
    e = _engine(1, 6)
    layers = [(g["body__0__weight"], g["body__0__bias"]), (g["body__4__weight"], g["body__4__bias"])]
    ln = [(g["body__1__weight"], g["body__1__bias"]), (g["body__5__weight"], g["body__5__bias"])]
    e.set_net_mlp(layers, ln, g["output__weight"], g["output__bias"])
    y = e.net_forward(g["queries"])
    assert np.abs(y - g["outputs"]).max() <= ATOL
    # and relative to the (small, x0.01-initialised) output scale
    assert np.abs(y - g["outputs"]).max() <= 1e-4 * np.abs(g["outputs"]).max()


@pytest.mark.parametrize("d,f,hidden,layers_n,use_ln,rows", [
    (1, 6, 256, 2, True, 1472), (1, 6, 256, 2, True, 31), (1, 6, 256, 2, True, 33), (1, 4, 256, 2, True, 700),
    (2, 3, 256, 2, True, 513), (2, 6, 256, 2, True, 300), (1, 6, 256, 3, True, 257), (1, 6, 256, 4, True, 200),
    (1, 6, 256, 1, False, 100), (1, 5, 256, 2, False, 129),
    # more 64-row groups than CUs: the persistent (register-resident-weights) kernel loops, prefetching the next group
    (1, 6, 256, 2, True, 40001), (2, 3, 256, 2, True, 33000), (2, 6, 256, 2, False, 20000)])
def test_mlp_vs_float64_reference(d, f, hidden, layers_n, use_ln, rows):
    _check_mlp(d, f, hidden, layers_n, use_ln, rows)


def test_unsupported_net_shape_is_refused_loudly():
    """Only n_hidden = 256 nets run on the MFMA forward; anything else raises (the rela layer then falls back to the
    caller's TorchScript module on the GPU, never to a CPU path)."""
    from rebel_amd import capi

    e = _engine(1, 6)
    rng = np.random.default_rng(0)
    layers = [(rng.uniform(-1, 1, (128, e.Q)).astype(np.float32), np.zeros(128, np.float32))]
    with pytest.raises(capi.RebelError, match="not supported"):
        e.set_net_mlp(layers, None, rng.uniform(-1, 1, (e.H, 128)).astype(np.float32), np.zeros(e.H, np.float32))


def test_mlp_edge_batches():
    """Empty batch, a single row, one row short of / one row past a 64-row group, and exactly one group per CU + 1 row."""
    for rows in (0, 1, 63, 65, 64 * 256 + 1):
        _check_mlp(1, 6, 256, 2, True, rows)


def test_mlp_fallback_kernel(monkeypatch):
    """The feature-split kernel (RBL_MLP_TILE=3) is the fallback for shapes the resident kernel does not take
    (n_layers != 2): same tolerance on the default shape too."""
    monkeypatch.setenv("RBL_MLP_TILE", "3")
    _check_mlp(1, 6, 256, 2, True, 5000)


@pytest.mark.parametrize("d,f,use_ln,rows", [
    (1, 6, True, 1472), (1, 6, True, 1), (1, 6, True, 31), (1, 6, True, 33), (1, 6, True, 65), (1, 6, True, 32 * 256 + 1),
    (1, 6, True, 40001), (1, 6, False, 129), (1, 4, True, 700), (1, 5, True, 2500), (1, 5, False, 96)])
def test_mlp_pipelined_kernel(monkeypatch, d, f, use_ln, rows):
    """The software-pipelined 32x32x16 kernel (RBL_MLP_TILE=6, opt-in: net_pipe_kernel.hip): same tolerance, including
    batches shorter than its pipeline depth (1-3 tiles), ragged last tiles and more tiles than CUs."""
    monkeypatch.setenv("RBL_MLP_TILE", "6")
    _check_mlp(d, f, 256, 2, use_ln, rows)


def test_mlp_pipelined_kernel_vs_torch_cpu_golden(monkeypatch):
    monkeypatch.setenv("RBL_MLP_TILE", "6")
    test_net2_vs_torch_cpu_golden()


def _check_mlp(d, f, hidden, layers_n, use_ln, rows):
    e = _engine(d, f)
    rng = np.random.default_rng(hidden + rows)
    Q, H = e.Q, e.H
    layers, ln = [], [] if use_ln else None
    n_in = Q
    for _ in range(layers_n):
        layers.append((rng.uniform(-1, 1, (hidden, n_in)).astype(np.float32) / np.sqrt(n_in),
                       rng.uniform(-0.1, 0.1, hidden).astype(np.float32)))
        if use_ln:
            ln.append((rng.uniform(0.5, 1.5, hidden).astype(np.float32), rng.uniform(-0.2, 0.2, hidden).astype(np.float32)))
        n_in = hidden
    w_out = rng.uniform(-1, 1, (H, hidden)).astype(np.float32) / np.sqrt(hidden)
    b_out = rng.uniform(-0.1, 0.1, H).astype(np.float32)
    e.set_net_mlp(layers, ln, w_out, b_out)
    # query-like inputs: flags, one-hot, two probability vectors
    q = np.zeros((rows, Q), np.float32)
    q[:, 0] = rng.integers(0, 2, rows)
    q[:, 1] = rng.integers(0, 2, rows)
    q[np.arange(rows), 2 + rng.integers(0, e.A, rows)] = 1
    q[:, 2 + e.A:2 + e.A + H] = rng.dirichlet(np.ones(H), rows)
    q[:, 2 + e.A + H:] = rng.dirichlet(np.ones(H), rows)
    y = e.net_forward(q)
    assert y.shape == (rows, H)
    if rows == 0:
        return
    ref = _np_net(q, layers, ln, w_out, b_out)
    assert np.abs(ref).max() > 0.05  # O(0.1-1) outputs: the tolerance is meaningful
    assert np.abs(y - ref).max() <= ATOL, np.abs(y - ref).max()


def test_asymmetric_identity_layout():
    """A = identity-like weights with an asymmetric pattern catches row/col or k-permutation mistakes exactly."""
    e = _engine(1, 6)
    Q, H, hid = e.Q, e.H, 256
    w0 = np.zeros((hid, Q), np.float32)
    for i in range(hid):
        w0[i, i % Q] = 1.0 + i / 1024.0
    w1 = np.zeros((hid, hid), np.float32)
    for i in range(hid):
        w1[i, (7 * i + 3) % hid] = 1.0
    w_out = np.zeros((H, hid), np.float32)
    for i in range(H):
        w_out[i, 11 * i + 5] = 1.0
    z = np.zeros(hid, np.float32)
    layers = [(w0, z), (w1, z)]
    e.set_net_mlp(layers, None, w_out, np.arange(H, dtype=np.float32))
    q = np.random.default_rng(0).uniform(0.5, 2.0, (70, Q)).astype(np.float32)
    ref = _np_net(q, layers, None, w_out, np.arange(H, dtype=np.float32))
    assert np.abs(e.net_forward(q) - ref).max() <= 1e-5
