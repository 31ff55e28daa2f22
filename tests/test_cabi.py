"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol the header declares,
fails loudly (never falls back to a CPU path) when there is no device, and the host-only entry points work."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "arkblst_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(pkg):
    L = pkg.load_library()
    syms = _declared_symbols()
    assert len(syms) >= 16
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/arkblst_amd.h but not exported"


def _exported(path):
    import subprocess

    out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
    return {l.split()[-1] for l in out.splitlines() if " T " in l}


def test_product_library_has_no_test_hooks(pkg):
    """mi_test_* entry points exist only in the test build (csrc/test_hooks.h); the product exports exactly the header"""
    prod = {s for s in _exported(pkg.lib_path()) if s.startswith("mi_")}
    assert prod == set(_declared_symbols()), prod ^ set(_declared_symbols())
    test = {s for s in _exported(pkg.lib_path(True)) if s.startswith("mi_")}
    assert test - prod == {"mi_test_fp_op", "mi_test_set_pairing", "mi_test_set_max_part", "mi_test_fail_allocs"}


def test_layout_sizes_match_reference_types(pkg):
    # blst_p1_affine 96, blst_p1 144, blst_p2_affine 192, blst_p2 288 (src/gpu.rs:69-71)
    from ark_blst_amd import binding as b

    assert (b.G1_AFF, b.G1_JAC, b.G2_AFF, b.G2_JAC) == (96, 144, 192, 288)
    assert C.sizeof(b.Profile) == 11 * 8 + 2 * 4 + 2 * 8 + 2 * 4


def test_strerror_and_invalid_args(pkg):
    L = pkg.load_library()
    assert L.mi_msm_strerror(0) == b"ok"
    assert L.mi_msm_strerror(-2) == b"no usable HIP device"
    assert L.mi_msm_init(None, None, 1) == -1  # MI_E_INVALID


def test_no_device_fails_loudly(pkg):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.MsmError) as e:
        pkg.Context([0])
    assert e.value.code == -2  # MI_E_NO_DEVICE: no silent CPU path


def test_host_fold_of_partials_matches_oracle(pkg, co, o):
    """mi_g1_sum / mi_g2_sum: the deterministic fold that follows the all-gather of per-GPU partial sums."""
    for group, F, gen in (("g1", o.F1, o.G1_GEN), ("g2", o.F2, o.G2_GEN)):
        ks = [5, 7, 11, 0, 123456789]
        parts = [o.jac_to_bytes(F, o.scalar_mul(F, gen, k)) for k in ks]
        # add a non-trivially scaled Jacobian representative: use the C oracle's Pippenger output as one partial
        aff = 96 if group == "g1" else 192
        bases = co.gen_bases(group, 5, 8, 1)
        sc = co.gen_scalars(6, 8)
        parts.append(co.msm(group, bases, sc, 8, 0, 1))
        got = (pkg.g1_sum if group == "g1" else pkg.g2_sum)(parts)
        want = co.sum_jac(group, b"".join(parts), len(parts))
        assert co.to_affine(group, got) == co.to_affine(group, want)
        # P + P (doubling inside the fold) and P + (-P)
        p = parts[1]
        dbl = (pkg.g1_sum if group == "g1" else pkg.g2_sum)([p, p])
        assert co.to_affine(group, dbl) == o.affine_to_bytes(F, o.scalar_mul(F, gen, 14))
        assert len(got) == (144 if group == "g1" else 288) and aff


def test_reference_style_length_mismatch_error(pkg):
    """VariableBaseMSM::msm returns Err(min(len)) on a length mismatch (arkworks default)."""
    with pytest.raises(pkg.msm.MsmErr) as e:
        pkg.G1Projective.msm(bytes(96 * 3), bytes(32 * 2))
    assert e.value.value == 2
