"""Compile-time resource guard for the two hot kernels (CPU only: hipcc cross-compiles gfx950 without a GPU).  A scratch spill
in the resident value-net kernel is a memory round trip inside its GEMM phases (HISTORY.md 3.2: it cost 9 k cycles per group
once), and the one-wavefront CFR kernel needs <= 128 VGPRs for its four waves per SIMD -- both were lost and recovered more than
once while the kernels were being changed, so the build checks them."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rebel_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _resources(src, extra=()):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I../../include", "-Wno-unused-result", *extra,
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull]
    r = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    out, name = {}, None
    for line in r.stdout.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
        for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and name:
                out[name][key] = int(m.group(1))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_resident_net_kernel_instantiations_that_are_launched_do_not_spill():
    res = _resources("net_resident_kernel.hip")
    seen = 0
    for name, r in res.items():
        m = re.search(r"mlp_resident_kernelILi(\d)ELb([01])ELi(\d)ELi(\d)ELi(\d)E", name)
        if not m:
            continue
        k0c, ln, notv, prod, nh = (int(x) for x in m.groups())
        seen += 1
        # shapes no game reaches: one or two input chunks (n_in <= 64: at most 2 dice x 4 faces, 16 hands) with more than one
        # output tile (> 16 hands)
        unreachable = k0c <= 2 and notv >= 3
        assert r["vgprs"] <= 256
        if not unreachable:
            assert r["scratch"] == 0, (name, r)
    # 4 input-chunk counts x (LayerNorm on / off) x 3 output variants, + the two half_inference modes; + two hidden layers
    # (n_layers = 3) for one / two input chunks and one output tile
    assert seen >= 56


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_resident_net_kernel_has_no_flat_loads(tmp_path):
    """A pointer that passes through an asm operand comes back generic and hipcc emits FLAT loads for it; those count on
    lgkmcnt too, so every LDS wait behind them also waits for an L2 round trip (rounds 2-4 shipped the per-group weight
    fetches that way; the streamed k-steps of the 2 dice x 6 faces kernel sat inside its hidden GEMM)."""
    out = tmp_path / "net_resident.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I../../include", "-Wno-unused-result", "-S",
           "--cuda-device-only", "net_resident_kernel.hip", "-o", str(out)]
    r = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    text = out.read_text()
    assert "v_mfma_f32_16x16x32" in text
    assert "flat_load" not in text and "flat_store" not in text


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src", ["selfplay_kernels.hip", "cfr_kernels.hip", "cfr_wave_kernel.hip", "cfr_flat_kernel.hip"])
def test_parity_kernels_have_no_flat_loads(tmp_path, src):
    """Round 6: hipcc 7.2 MISCOMPILED `flag ? a.A : a.lane_shape[i]` in sp_order_kernel -- a select between a kernel-argument field
    and a loaded value became ONE flat load through a selected pointer (&kernarg.A or &lane_shape[i]) and the condition was lost, so
    every lane sorted under the same key (root de-duplication then sized the 2 dice x 6 faces launch segments by whatever lane came
    first and hung).  The signature of that code shape is a FLAT load in a kernel whose every pointer is a global one: none may
    appear in the bit-exact kernels."""
    out = tmp_path / (src + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I../../include", "-Wno-unused-result", "-ffp-contract=off",
           "-S", "--cuda-device-only", src, "-o", str(out)]
    r = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    text = out.read_text()
    assert "s_endpgm" in text and "flat_load" not in text and "flat_store" not in text


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_wave_cfr_kernel_keeps_four_waves_per_simd_for_the_one_die_games():
    res = _resources("cfr_wave_kernel.hip", ("-ffp-contract=off",))
    by_h = {}
    for name, r in res.items():
        m = re.search(r"cfr_wave_kernelILi(\d+)ELi(\d+)ELi(\d)ELi(\d)E", name)
        if m:
            by_h[(int(m.group(3)), int(m.group(4)))] = r
    assert set(by_h) == {(1, 4), (1, 5), (1, 6), (2, 3)}
    for game, r in by_h.items():
        assert r["scratch"] == 0, (game, r)
    # 1 die x 6 faces (the headline): its LDS image allows 18 lanes per CU (round 5), i.e. five waves on some SIMDs: <= 96 VGPRs;
    # 2 dice x 3 faces: 12 lanes per CU need <= 128
    assert by_h[(1, 6)]["vgprs"] <= 96 and by_h[(1, 4)]["vgprs"] <= 96 and by_h[(2, 3)]["vgprs"] <= 128, by_h


def test_wave_cfr_kernel_lds_image_keeps_its_lanes_per_cu():
    """The one-wavefront CFR kernel is LDS-limited: gfx950 allocates LDS in 1 280-byte granules (measured: the launch time
    steps exactly there, profiles/r05_cfr_wave_occupancy_sensitivity.txt) and every two lanes per CU are worth ~3 % of the
    launch.  Root subgame of the headline game: <= 7 granules (18 lanes per CU); 2 dice x 3 faces: <= 10 granules (12)."""
    import ctypes

    lib = os.path.join(ROOT, "rebel_amd", "librebel_hip.so")
    if not os.path.exists(lib):
        pytest.skip("librebel_hip.so not built")
    f = getattr(ctypes.CDLL(lib), "_ZN3rbl18cfr_wave_lds_bytesEiiiiiiii")  # rbl::cfr_wave_lds_bytes(N, NI, H, L, T, faces, lo_d, lo_p)
    f.restype = ctypes.c_size_t
    # depth-2 root subgame of a 13-action game: 1 + 12 + 78 nodes, 66 pseudo-leaves, 12 terminals, 13 nodes with children
    assert f(91, 13, 6, 66, 12, 6, 13, 1) <= 7 * 1280
    assert f(91, 13, 9, 66, 12, 3, 13, 1) <= 10 * 1280
