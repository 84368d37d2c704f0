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


def test_class_default_depth_runs_on_the_resident_kernel():
    """Net2's class default is n_layers = 3 (cfvpy/models.py:73).  The register-resident kernel takes it (two hidden layers,
    each half resident, half streamed) for the one-die games and 2 dice x 3 faces; results within the same 1e-5."""
    for d, f in ((1, 6), (1, 4), (2, 3)):
        e, err = _check_mlp(d, f, 256, 3, True, 40001, return_engine=True)
        assert e.stats()["net_kernel"] == 5
        assert err <= 2e-6
    # 2 dice x 6 faces (four input chunks) and half_inference arithmetic at this depth: the engine says which kernel it picked
    e, _ = _check_mlp(2, 6, 256, 3, True, 3000, return_engine=True)
    assert e.stats()["net_kernel"] == 3


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
    (n_layers > 3, or n_layers = 3 at 2 dice x 6 faces): same tolerance on the default shape too."""
    monkeypatch.setenv("RBL_MLP_TILE", "3")
    _check_mlp(1, 6, 256, 2, True, 5000)


def _check_mlp(d, f, hidden, layers_n, use_ln, rows, precision=0, atol=ATOL, return_engine=False):
    e = _engine(d, f)
    e.set_net_precision(precision)
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
    assert np.abs(y - ref).max() <= atol, np.abs(y - ref).max()
    if return_engine:
        return e, np.abs(y - ref).max()
    return np.abs(y - ref).max()


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


# ---------------------------------------------------------------------------------------------- half_inference (round 4)
def _half_case(d, f, rows, seed, out_scale=1.0, n_layers=2):
    """A Net2 whose parameters are f16-representable (what model.half() leaves), query-like inputs, and the outputs of
    (a) float64 arithmetic on those weights, (b) the SAME module as a half torch module on the GPU (the reference's
    half_inference path: cfvpy/selfplay.py:42-43, 211; rela/model_locker.h:85-95)."""
    import torch

    from rebel_amd.models import Net2, mlp_weights_from_state_dict

    torch.manual_seed(seed)
    net = Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=n_layers).eval()
    with torch.no_grad():
        for prm in net.parameters():
            prm.mul_(1.0 + 0.3 * torch.rand_like(prm))  # LayerNorm weights / biases away from exactly 1 / 0
        net.output.weight *= out_scale / 0.01 * 0.1  # Net2 initialises the output layer x0.01: bring the outputs to O(0.1-1)
        net.output.bias *= out_scale / 0.01 * 0.1
    half = net.half()
    sd32 = {k: v.float().cpu() for k, v in half.state_dict().items()}
    A, H = 2 * d * f + 1, f ** d
    Q = 2 + A + 2 * H
    rng = np.random.default_rng(seed)
    q = np.zeros((rows, Q), np.float32)
    q[:, 0] = rng.integers(0, 2, rows)
    q[:, 1] = rng.integers(0, 2, rows)
    q[np.arange(rows), 2 + rng.integers(0, A, rows)] = 1
    q[:, 2 + A:2 + A + H] = rng.dirichlet(np.ones(H), rows)
    q[:, 2 + A + H:] = rng.dirichlet(np.ones(H), rows)
    layers, ln, w_out, b_out = mlp_weights_from_state_dict(sd32)
    ref64 = _np_net(q, layers, ln, w_out, b_out)
    with torch.no_grad():
        y_half = half.to("cuda:0")(torch.from_numpy(q).to("cuda:0").half()).float().cpu().numpy()
    return (layers, ln, w_out, b_out), q, ref64, y_half


# n_layers = 3 is Net2's CLASS DEFAULT (cfvpy/models.py:73): a half TorchScript module of that depth reaches precision 2 on the
# NH = 2 instantiations of the resident kernel through rela's apply_mlp (that combination had no parity case)
@pytest.mark.parametrize("d,f,rows,n_layers", [(1, 6, 4000, 2), (1, 4, 1500, 2), (2, 3, 3000, 2), (2, 6, 2500, 2), (1, 6, 65, 2),
                                               (1, 6, 4000, 3), (1, 4, 1500, 3), (2, 3, 3000, 3), (1, 6, 65, 3)])
def test_half_inference_modes_vs_the_half_torch_module(d, f, rows, n_layers):
    """rbl_engine_set_net_precision (VERDICT r3 missing #3).  Against float64 arithmetic on the half model's weights:
    mode 0 keeps the f32-parity bar; mode 1 (activations rounded to f16, two products) and mode 2 (activations and packed
    weights rounded to f16, one product: the reference's half_inference semantics with f32 accumulation / LayerNorm / GELU)
    are each AT LEAST as accurate as the half torch module itself run on this GPU -- the reference's own arithmetic for
    this setting -- in maximum and in mean absolute error."""
    weights, q, ref64, y_half = _half_case(d, f, rows, seed=3 + rows, n_layers=n_layers)
    assert np.abs(ref64).max() > 0.05
    err_torch = np.abs(y_half - ref64)
    errs = {}
    for mode in (0, 1, 2):
        e = _engine(d, f)
        e.set_net_precision(mode)
        e.set_net_mlp(*weights)
        y = e.net_forward(q)
        st = e.stats()
        assert st["net_products"] == 3 - mode and st["net_kernel"] == 5  # the register-resident kernel, both depths
        errs[mode] = np.abs(y - ref64)
    print(f"half_inference {d}dx{f}f n_layers {n_layers}: max|err| vs float64 -- torch half module {err_torch.max():.3e} (mean {err_torch.mean():.3e}); "
          + "; ".join(f"mode {m}: {v.max():.3e} (mean {v.mean():.3e})" for m, v in errs.items()))
    assert errs[0].max() <= ATOL
    for mode in (1, 2):
        assert errs[mode].max() <= err_torch.max(), (mode, errs[mode].max(), err_torch.max())
        assert errs[mode].mean() <= err_torch.mean(), (mode, errs[mode].mean(), err_torch.mean())
    assert errs[1].mean() <= errs[2].mean() * 1.05  # keeping the weights' low halves does not hurt


def test_half_inference_modes_are_refused_where_they_do_not_exist():
    from rebel_amd import capi

    e = _engine(1, 6)
    rng = np.random.default_rng(0)
    with pytest.raises(capi.RebelError, match="mode must be"):
        e.set_net_precision(3)
    e.set_net_precision(2)
    layers = [(rng.uniform(-1, 1, (256, e.Q)).astype(np.float32), np.zeros(256, np.float32)),
              (rng.uniform(-1, 1, (256, 256)).astype(np.float32), np.zeros(256, np.float32))]
    with pytest.raises(capi.RebelError, match="half_inference"):  # no LayerNorm
        e.set_net_mlp(layers, None, rng.uniform(-1, 1, (e.H, 256)).astype(np.float32), np.zeros(e.H, np.float32))
    e.set_net_precision(0)
    e.set_net_mlp(layers, None, rng.uniform(-1, 1, (e.H, 256)).astype(np.float32), np.zeros(e.H, np.float32))


def test_a_refused_net_leaves_the_previous_one_intact():
    """set_net_mlp used to reject an unsupported precision / net combination only after it had replaced
    the weight blob, reset the kernel descriptor and switched the query layout -- a live engine was left wedged ('unknown
    kernel variant' on every later forward).  Every refusal now happens before the engine is touched: the forward of the
    net that was there still gives the same answers bit for bit, and a retry in mode 0 works."""
    from rebel_amd import capi

    weights, q, ref64, _ = _half_case(1, 6, 500, seed=11)
    e = _engine(1, 6)
    e.set_net_mlp(*weights)
    before = e.net_forward(q)
    assert np.abs(before - ref64).max() <= ATOL
    rng = np.random.default_rng(1)
    no_ln = [(rng.uniform(-1, 1, (256, e.Q)).astype(np.float32), np.zeros(256, np.float32)),
             (rng.uniform(-1, 1, (256, 256)).astype(np.float32), np.zeros(256, np.float32))]
    w_out, b_out = rng.uniform(-1, 1, (e.H, 256)).astype(np.float32), np.zeros(e.H, np.float32)
    e.set_net_precision(2)
    with pytest.raises(capi.RebelError, match="half_inference"):
        e.set_net_mlp(no_ln, None, w_out, b_out)
    assert np.array_equal(e.net_forward(q), before)  # same blob, same descriptor, same layout
    assert e.stats()["net_products"] == 3
    e.set_net_precision(0)
    e.set_net_mlp(no_ln, None, w_out, b_out)  # the retry the message suggests
    y = e.net_forward(q)
    assert np.abs(y - _np_net(q, no_ln, None, w_out, b_out)).max() <= ATOL * max(1.0, np.abs(y).max())


def test_half_inference_edge_batches():
    for rows in (1, 63, 64 * 256 + 1):
        for mode in (1, 2):
            _check_mlp(1, 6, 256, 2, True, rows, precision=mode, atol=3e-3)
