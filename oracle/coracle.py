"""CPU ORACLE (test infrastructure only) — ctypes loader for oracle/libmsm_oracle.so (see msm_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

G1_AFF, G1_JAC, G2_AFF, G2_JAC = 96, 144, 192, 288


def build() -> str:
    """Compile the C restatement (gcc only; seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return os.path.join(_HERE, "libmsm_oracle.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmsm_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        u8p, sz, i32, u64, u32 = C.c_char_p, C.c_size_t, C.c_int, C.c_uint64, C.c_uint
        for g in ("g1", "g2"):
            getattr(L, f"orc_{g}_msm").argtypes = [u8p, u8p, sz, i32, i32, u8p]
            getattr(L, f"orc_{g}_msm").restype = i32
            getattr(L, f"orc_{g}_msm_naive").argtypes = [u8p, u8p, sz, i32, u8p]
            getattr(L, f"orc_{g}_msm_naive").restype = i32
            getattr(L, f"orc_{g}_to_affine").argtypes = [u8p, u8p]
            getattr(L, f"orc_{g}_sum_jac").argtypes = [u8p, sz, u8p]
            getattr(L, f"orc_{g}_fold_windows").argtypes = [u8p, u32, u32, u8p]
            getattr(L, f"orc_{g}_mul_gen").argtypes = [u8p, u8p]
            getattr(L, f"orc_{g}_gen_bases").argtypes = [u64, sz, i32, u8p]
            getattr(L, f"orc_{g}_on_curve").argtypes = [u8p]
            getattr(L, f"orc_{g}_on_curve").restype = i32
            getattr(L, f"orc_{g}_normalize_batch").argtypes = [u8p, sz, i32, u8p]
        L.orc_g1_deserialize_batch.argtypes = [u8p, sz, i32, i32, i32, i32, u8p, u8p]
        L.orc_g1_deserialize_batch.restype = i32
        L.orc_g2_deserialize_batch.argtypes = [u8p, sz, i32, i32, i32, i32, u8p, u8p]
        L.orc_g2_deserialize_batch.restype = i32
        L.orc_gen_scalars.argtypes = [u64, sz, i32, u8p]
        L.orc_gen_dlogs.argtypes = [u64, sz, u8p]
        L.orc_dot_mod_r.argtypes = [u8p, u64, sz, u8p]
        for f in ("orc_fp_mul", "orc_fp_add", "orc_fp_sub"):
            getattr(L, f).argtypes = [u8p, u8p, u8p, sz]
        L.orc_fp_to_mont.argtypes = [u8p, u8p]
        L.orc_fp_from_mont.argtypes = [u8p, u8p]
        L.orc_fr_from_mont.argtypes = [u8p, u8p, sz]
        L.orc_fr_to_mont.argtypes = [u8p, u8p, sz]
        L.orc_selfcheck.restype = i32
        _LIB = L
    return _LIB


def _sizes(group: str):
    return (G1_AFF, G1_JAC) if group == "g1" else (G2_AFF, G2_JAC)


def selfcheck() -> int:
    return lib().orc_selfcheck()


def msm(group: str, bases: bytes, scalars: bytes, n: int, scalar_fmt: int = 0, nthreads: int = 1) -> bytes:
    """Pippenger restatement. Returns the Jacobian result bytes (144 / 288)."""
    aff, jac = _sizes(group)
    assert len(bases) >= aff * n and len(scalars) >= 32 * n
    out = C.create_string_buffer(jac)
    rc = getattr(lib(), f"orc_{group}_msm")(bases, scalars, n, scalar_fmt, nthreads, out)
    assert rc == 0
    return out.raw


def msm_naive(group: str, bases: bytes, scalars: bytes, n: int, scalar_fmt: int = 0) -> bytes:
    aff, jac = _sizes(group)
    out = C.create_string_buffer(jac)
    getattr(lib(), f"orc_{group}_msm_naive")(bases, scalars, n, scalar_fmt, out)
    return out.raw


def to_affine(group: str, jac_bytes: bytes) -> bytes:
    """Canonical comparison form: fully reduced Montgomery limbs of affine (x, y); infinity = zeros."""
    aff, jac = _sizes(group)
    assert len(jac_bytes) == jac
    out = C.create_string_buffer(aff)
    getattr(lib(), f"orc_{group}_to_affine")(jac_bytes, out)
    return out.raw


def sum_jac(group: str, pts: bytes, n: int) -> bytes:
    aff, jac = _sizes(group)
    out = C.create_string_buffer(jac)
    getattr(lib(), f"orc_{group}_sum_jac")(pts, n, out)
    return out.raw


def fold_windows(group: str, wins: bytes, nwin: int, c: int) -> bytes:
    aff, jac = _sizes(group)
    out = C.create_string_buffer(jac)
    getattr(lib(), f"orc_{group}_fold_windows")(wins, nwin, c, out)
    return out.raw


def mul_gen(group: str, k: int) -> bytes:
    aff, jac = _sizes(group)
    out = C.create_string_buffer(aff)
    getattr(lib(), f"orc_{group}_mul_gen")(int(k).to_bytes(32, "little"), out)
    return out.raw


def gen_bases(group: str, seed: int, n: int, nthreads: int = 1) -> bytes:
    aff, jac = _sizes(group)
    out = C.create_string_buffer(aff * n)
    getattr(lib(), f"orc_{group}_gen_bases")(seed, n, nthreads, out)
    return out.raw


def gen_scalars(seed: int, n: int, mont: bool = False) -> bytes:
    out = C.create_string_buffer(32 * n)
    lib().orc_gen_scalars(seed, n, 1 if mont else 0, out)
    return out.raw


def dlog_expected(group: str, scalars_canon: bytes, seed_bases: int, n: int) -> bytes:
    """Closed form (sum_i s_i k_i mod r) * G as canonical affine bytes."""
    dot = C.create_string_buffer(32)
    lib().orc_dot_mod_r(scalars_canon, seed_bases, n, dot)
    return mul_gen(group, int.from_bytes(dot.raw, "little"))


def normalize_batch(group: str, jac: bytes, nthreads: int = 1) -> bytes:
    """CurveGroup::normalize_batch (src/g1.rs:537-543): Jacobian -> affine with one inversion per thread slice."""
    aff, jsz = _sizes(group)
    n = len(jac) // jsz
    assert len(jac) == n * jsz
    out = C.create_string_buffer(aff * n)
    getattr(lib(), f"orc_{group}_normalize_batch")(jac, n, nthreads, out)
    return out.raw


def g1_deserialize_batch(data: bytes, compressed: bool, validate: bool, subgroup_mode: int = 0, nthreads: int = 1):
    """G1 point decoding + Valid::check (src/g1.rs:386-431).  subgroup_mode 0: [r] P == infinity (the definition, the checker);
    1: the endomorphism test (the timed baseline).  Returns (affine bytes, status bytes)."""
    size = 48 if compressed else 96
    n = len(data) // size
    assert len(data) == n * size
    out, st = C.create_string_buffer(G1_AFF * n), C.create_string_buffer(n)
    rc = lib().orc_g1_deserialize_batch(data, n, int(compressed), int(validate), subgroup_mode, nthreads, out, st)
    assert rc == 0
    return out.raw, st.raw


def g2_deserialize_batch(data: bytes, compressed: bool, validate: bool, subgroup_mode: int = 0, nthreads: int = 1):
    """(affine bytes, status bytes) — /root/reference/src/g2.rs:366-411 restated in oracle/msm_oracle.c (orc_g2_deserialize_batch)"""
    size = 96 if compressed else 192
    n = len(data) // size
    out = C.create_string_buffer(192 * n)
    st = C.create_string_buffer(n)
    rc = lib().orc_g2_deserialize_batch(data, n, int(compressed), int(validate), subgroup_mode, nthreads, out, st)
    if rc != 0:
        raise RuntimeError("orc_g2_deserialize_batch: the psi constants failed their self-check")
    return out.raw, st.raw


def fp_mul(a: bytes, b: bytes) -> bytes:
    n = len(a) // 48
    out = C.create_string_buffer(48 * n)
    lib().orc_fp_mul(a, b, out, n)
    return out.raw


def fr_from_mont(a: bytes) -> bytes:
    n = len(a) // 32
    out = C.create_string_buffer(32 * n)
    lib().orc_fr_from_mont(a, out, n)
    return out.raw


def fr_to_mont(a: bytes) -> bytes:
    n = len(a) // 32
    out = C.create_string_buffer(32 * n)
    lib().orc_fr_to_mont(a, out, n)
    return out.raw


# ---- pairing (oracle/pairing_oracle.c) -------------------------------------------------------------------------
_PLIB = None


def plib():
    global _PLIB
    if _PLIB is None:
        path = os.path.join(_HERE, "libpairing_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_multi_miller_loop.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.c_char_p]
        L.orc_fp12_pow.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p]
        L.orc_fp12_mul.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        _PLIB = L
    return _PLIB


def multi_miller_loop(g1: bytes, g2: bytes, nthreads: int = 1) -> bytes:
    n = len(g1) // G1_AFF
    assert len(g1) == n * G1_AFF and len(g2) == n * G2_AFF
    out = C.create_string_buffer(576)
    plib().orc_multi_miller_loop(g1, g2, n, nthreads, out)
    return out.raw


def final_exponentiation(f: bytes) -> bytes:
    from . import pairing as pr   # the exponent 3 (p^12 - 1) / r as an integer

    e = pr.FINAL_EXP.to_bytes((pr.FINAL_EXP.bit_length() + 7) // 8, "little")
    out = C.create_string_buffer(576)
    plib().orc_fp12_pow(f, e, len(e), out)
    return out.raw


def multi_pairing(g1: bytes, g2: bytes, nthreads: int = 1) -> bytes:
    return final_exponentiation(multi_miller_loop(g1, g2, nthreads))
