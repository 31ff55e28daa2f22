"""CPU tests of the pairing row: the oracle against its own properties and the golden vectors, the
shipped tower / Miller loop / final exponentiation compiled for the host (ark-blst_amd/csrc/pairing.cuh is generic
host+device code) against the oracle, the machine check of its lazy-reduction bounds, and the host-only C-ABI entry
point `mi_final_exponentiation`."""
import ctypes as C
import json
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def pr():
    from oracle import pairing

    return pairing


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(HERE, "golden", "pairing_vectors.json")))


@pytest.fixture(scope="module")
def hp():
    src = os.path.join(HERE, "host", "pairing_host_check.cpp")
    so = os.path.join(HERE, "host", "libpairing_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    return C.CDLL(so)


def test_oracle_selfcheck(pr):
    """bilinearity (the reference's own test, src/pairing.rs:92-101), non-degeneracy, order r, blst's exponent identity"""
    pr.selfcheck()


def test_oracle_reproduces_golden(pr, o, gold):
    for c in gold["multi_pairing"]:
        if c["n"] > 2:
            continue   # keep the CPU suite short; the GPU suite covers every case
        g1, g2 = bytes.fromhex(c["g1"]), bytes.fromhex(c["g2"])
        ps = [o.affine_from_bytes(o.F1, g1[96 * i:96 * i + 96]) for i in range(c["n"])]
        qs = [o.affine_from_bytes(o.F2, g2[192 * i:192 * i + 192]) for i in range(c["n"])]
        assert pr.fp12_to_bytes(pr.final_exponentiation(pr.multi_miller_loop(ps, qs))).hex() == c["gt"], c["name"]


def test_c_oracle_vs_python_oracle_and_golden(co, pr, o, gold):
    """oracle/pairing_oracle.c (fast checker + timed CPU baseline) against oracle/pairing.py and the frozen vectors"""
    for c in gold["multi_pairing"]:
        assert co.multi_pairing(bytes.fromhex(c["g1"]), bytes.fromhex(c["g2"]), 2).hex() == c["gt"], c["name"]
    g = gold["final_exponentiation"]
    assert co.final_exponentiation(bytes.fromhex(g["f"])).hex() == g["out"]
    g1 = co.gen_bases("g1", 31, 3, 1)
    g2 = co.gen_bases("g2", 32, 3, 1)
    ps = [o.affine_from_bytes(o.F1, g1[96 * i:96 * i + 96]) for i in range(3)]
    qs = [o.affine_from_bytes(o.F2, g2[192 * i:192 * i + 192]) for i in range(3)]
    assert co.multi_miller_loop(g1, g2, 3) == pr.fp12_to_bytes(pr.multi_miller_loop(ps, qs))   # same algorithm: equal before the exponentiation too


def test_lazy_reduction_bounds_of_shipped_code():
    exe = os.path.join(HERE, "host", "pairing_bounds")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(HERE, "host", "pairing_bounds.cpp")])
    out = subprocess.check_output([exe]).decode()
    assert "pairing bounds OK" in out


def test_host_build_tower_vs_oracle(hp, pr, o, gold):
    rnd = random.Random(11)
    out = C.create_string_buffer(576)
    for _ in range(4):
        a = [(rnd.randrange(o.P), rnd.randrange(o.P)) for _ in range(6)]
        b = [(rnd.randrange(o.P), rnd.randrange(o.P)) for _ in range(6)]
        hp.hp_fp12_mul(pr.fp12_to_bytes(a), pr.fp12_to_bytes(b), out)
        assert out.raw == pr.fp12_to_bytes(pr.fp12_mul(a, b))
        hp.hp_fp12_sqr(pr.fp12_to_bytes(a), out)
        assert out.raw == pr.fp12_to_bytes(pr.fp12_mul(a, a))
        hp.hp_fp12_inv(pr.fp12_to_bytes(a), out)
        assert pr.fp12_eq(pr.fp12_mul(pr.fp12_from_bytes(out.raw), a), pr.FP12_ONE)
        hp.hp_fp12_frob(pr.fp12_to_bytes(a), out)
        assert out.raw == pr.fp12_to_bytes(pr.fp12_pow(a, o.P))
    # sparse / edge operands: zero halves, the one, p-1 coefficients
    edge = [[(0, 0)] * 6, list(pr.FP12_ONE), [(o.P - 1, o.P - 1)] * 6, [(0, 0), (5, 0), (0, 0), (0, 0), (0, 7), (0, 0)]]
    for a in edge:
        for b in edge:
            hp.hp_fp12_mul(pr.fp12_to_bytes(a), pr.fp12_to_bytes(b), out)
            assert out.raw == pr.fp12_to_bytes(pr.fp12_mul(a, b))
    g = gold["fp12_mul"]
    hp.hp_fp12_mul(bytes.fromhex(g["a"]), bytes.fromhex(g["b"]), out)
    assert out.raw.hex() == g["ab"]


def test_host_build_pairing_vs_golden(hp, gold):
    out = C.create_string_buffer(576)
    for c in gold["multi_pairing"]:
        if c["n"] != 1:
            continue
        hp.hp_miller_loop(bytes.fromhex(c["g1"]), bytes.fromhex(c["g2"]), out)
        hp.hp_final_exp(out.raw, out)
        assert out.raw.hex() == c["gt"], c["name"]


def test_cabi_final_exponentiation(pkg, pr, o, gold):
    """host-only entry point: works without a device"""
    g = gold["final_exponentiation"]
    assert pkg.final_exponentiation(bytes.fromhex(g["f"])).hex() == g["out"]
    one = pr.fp12_to_bytes(pr.FP12_ONE)
    assert pkg.final_exponentiation(one) == one
    L = pkg.load_library()
    assert L.mi_final_exponentiation(None, None) == -1   # MI_E_INVALID
    assert L.mi_multi_pairing(None, None, None, 0, None) == -1
