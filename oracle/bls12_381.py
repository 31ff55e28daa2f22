"""CPU ORACLE (test infrastructure only) — Python big-int restatement of the BLS12-381 MSM path.

NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the shipped path (ark-blst_amd/) never does.

What it restates (reference = nikkolasg/ark-blst, paths relative to /root/reference):
  * `<G1Projective as VariableBaseMSM>::msm`  src/g1.rs:602-619 (CPU) / 621-632 (GPU)  ->  msm_g1()
  * `<G2Projective as VariableBaseMSM>::msm`  src/g2.rs:582-599 / 601-612              ->  msm_g2()
  * the GPU driver's scalar convention (256-bit little-endian integer, src/gpu.rs:126-223)
  * `Scalar::into_bigint` (Montgomery -> canonical), src/scalar.rs:450-463,503-505     ->  fr_from_mont()
  * in-memory layouts the reference ships raw to the device (src/gpu.rs:149-156,185-186):
    blst_fp = 6 x u64 LE limbs, Montgomery R = 2^384; blst_fr = 4 x u64, R = 2^256;
    affine = (x, y), all-zero = infinity; Jacobian = (X, Y, Z), Z = 0 = infinity.

The arithmetic itself lives in third-party crates that are ABSENT from /root/reference
(blst =0.3.10, blstrs ^0.6.1 [git branch, no lockfile], ec-gpu-gen 0.5.1 — Cargo.toml:16,22-24,58-62),
so this file restates the *published* algorithms (short-Weierstrass group law on y^2 = x^3 + 4 over
Fp and y^2 = x^3 + 4(1+u) over Fp2, Montgomery representation) and anchors on the reference's call
sites and embedded constants.

PINNING STATUS
  * Field / encoding layer: PINNED by the reference's own known-answer constants, checked in
    selfcheck(): Fp modulus (src/fp.rs:25-32), Fr modulus (src/scalar.rs:476-481), the Fp Montgomery
    KAT mont((p-1)/2) (src/fp.rs:714-721), G1 cofactor + COFACTOR_INV Montgomery limbs
    (src/g1.rs:42-51), G2 cofactor + COFACTOR_INV (src/g2.rs:45-63).
  * MSM results: **parity unpinned** — the reference holds no golden vector, fixture or fixed seed
    for MSM (its only MSM test is the randomized property msm == sum b_i*s_i, src/tests.rs:50-67),
    and the reference cannot be built here (Rust toolchain absent, un-vendored git dependencies).
    The contract is therefore the mathematical definition that property asserts, which msm_naive()
    implements directly and tests/golden/ freezes.
"""

from __future__ import annotations

# ----------------------------------------------------------------------------------------------
# constants (each verified in selfcheck())
# ----------------------------------------------------------------------------------------------
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R_ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
X_PARAM = -0xD201000000010000
H1 = 0x396C8C005555E1568C00AAAB0000AAAB

FP_R = (1 << 384) % P          # Montgomery radix for blst_fp
FR_R = (1 << 256) % R_ORDER    # Montgomery radix for blst_fr
FP_RINV = pow(FP_R, -1, P)
FR_RINV = pow(FR_R, -1, R_ORDER)

G1_X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1_Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
G2_X = (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
        0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E)
G2_Y = (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
        0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE)


# ----------------------------------------------------------------------------------------------
# field towers as tiny op-tables so G1 and G2 share the group law
# ----------------------------------------------------------------------------------------------
class _Fp:
    zero = 0
    one = 1
    b = 4  # y^2 = x^3 + 4
    nlimbs = 6

    @staticmethod
    def add(a, b): return (a + b) % P
    @staticmethod
    def sub(a, b): return (a - b) % P
    @staticmethod
    def mul(a, b): return (a * b) % P
    @staticmethod
    def neg(a): return (-a) % P
    @staticmethod
    def inv(a): return pow(a, -1, P)
    @staticmethod
    def is_zero(a): return a % P == 0
    @staticmethod
    def eq(a, b): return (a - b) % P == 0


class _Fp2:
    zero = (0, 0)
    one = (1, 0)
    b = (4, 4)  # y^2 = x^3 + 4(1+u)
    nlimbs = 12

    @staticmethod
    def add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
    @staticmethod
    def sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
    @staticmethod
    def mul(a, b):  # (a0 + a1 u)(b0 + b1 u), u^2 = -1
        return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
    @staticmethod
    def neg(a): return ((-a[0]) % P, (-a[1]) % P)
    @staticmethod
    def inv(a):
        n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
        return ((a[0] * n) % P, (-a[1] * n) % P)
    @staticmethod
    def is_zero(a): return a[0] % P == 0 and a[1] % P == 0
    @staticmethod
    def eq(a, b): return (a[0] - b[0]) % P == 0 and (a[1] - b[1]) % P == 0


F1, F2 = _Fp, _Fp2
INF = None  # affine point at infinity


def on_curve(F, pt) -> bool:
    if pt is INF:
        return True
    x, y = pt
    return F.eq(F.mul(y, y), F.add(F.mul(F.mul(x, x), x), F.b))


def aff_neg(F, p):
    return INF if p is INF else (p[0], F.neg(p[1]))


def aff_add(F, p, q):
    """Complete affine addition (handles infinity, doubling, inverse)."""
    if p is INF:
        return q
    if q is INF:
        return p
    x1, y1 = p
    x2, y2 = q
    if F.eq(x1, x2):
        if F.eq(y1, y2) and not F.is_zero(y1):
            num = F.mul(F.add(F.add(F.mul(x1, x1), F.mul(x1, x1)), F.mul(x1, x1)), F.one)
            lam = F.mul(num, F.inv(F.add(y1, y1)))
        else:
            return INF
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


# Jacobian ops (fast path for scalar multiplication; (X, Y, Z), Z = 0 <=> infinity)
def jac_from_aff(F, p):
    return (F.one, F.one, F.zero) if p is INF else (p[0], p[1], F.one)


def jac_to_aff(F, j):
    X, Y, Z = j
    if F.is_zero(Z):
        return INF
    zi = F.inv(Z)
    zi2 = F.mul(zi, zi)
    return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))


def jac_double(F, j):
    X, Y, Z = j
    if F.is_zero(Z) or F.is_zero(Y):
        return (F.one, F.one, F.zero)
    A = F.mul(X, X)
    B = F.mul(Y, Y)
    C = F.mul(B, B)
    t = F.add(X, B)
    D = F.sub(F.sub(F.mul(t, t), A), C)
    D = F.add(D, D)
    E = F.add(F.add(A, A), A)
    Fq = F.mul(E, E)
    X3 = F.sub(Fq, F.add(D, D))
    C8 = F.add(C, C); C8 = F.add(C8, C8); C8 = F.add(C8, C8)
    Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
    Z3 = F.mul(F.add(Y, Y), Z)
    return (X3, Y3, Z3)


def jac_add(F, p, q):
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    if F.is_zero(Z1):
        return q
    if F.is_zero(Z2):
        return p
    Z1Z1 = F.mul(Z1, Z1)
    Z2Z2 = F.mul(Z2, Z2)
    U1 = F.mul(X1, Z2Z2)
    U2 = F.mul(X2, Z1Z1)
    S1 = F.mul(Y1, F.mul(Z2, Z2Z2))
    S2 = F.mul(Y2, F.mul(Z1, Z1Z1))
    if F.eq(U1, U2):
        if F.eq(S1, S2):
            return jac_double(F, p)
        return (F.one, F.one, F.zero)
    H = F.sub(U2, U1)
    Rr = F.sub(S2, S1)
    HH = F.mul(H, H)
    HHH = F.mul(H, HH)
    V = F.mul(U1, HH)
    X3 = F.sub(F.sub(F.mul(Rr, Rr), HHH), F.add(V, V))
    Y3 = F.sub(F.mul(Rr, F.sub(V, X3)), F.mul(S1, HHH))
    Z3 = F.mul(F.mul(Z1, Z2), H)
    return (X3, Y3, Z3)


def scalar_mul(F, p, k: int):
    """k * p by double-and-add (cf. bit-serial mul_bigint, src/g1.rs:331-341, 513-527)."""
    if p is INF or k == 0:
        return INF
    if k < 0:
        return scalar_mul(F, aff_neg(F, p), -k)
    acc = (F.one, F.one, F.zero)
    base = jac_from_aff(F, p)
    for bit in bin(k)[2:]:
        acc = jac_double(F, acc)
        if bit == "1":
            acc = jac_add(F, acc, base)
    return jac_to_aff(F, acc)


def msm_naive(F, bases, scalars):
    """Definition the reference's own MSM test asserts (src/tests.rs:57-67): sum_i s_i * b_i.
    Truncates to min(len) like the CPU impl (src/g1.rs:604-618 via blstrs multi_exp)."""
    n = min(len(bases), len(scalars))
    acc = (F.one, F.one, F.zero)
    for i in range(n):
        t = scalar_mul(F, bases[i], scalars[i] % R_ORDER)
        acc = jac_add(F, acc, jac_from_aff(F, t))
    return jac_to_aff(F, acc)


def msm_g1(bases, scalars):
    return msm_naive(F1, bases, scalars)


def msm_g2(bases, scalars):
    return msm_naive(F2, bases, scalars)


G1_GEN = (G1_X, G1_Y)
G2_GEN = (G2_X, G2_Y)


# ----------------------------------------------------------------------------------------------
# in-memory encodings (the device ABI)
# ----------------------------------------------------------------------------------------------
def fp_to_mont_bytes(a: int) -> bytes:
    return ((a * FP_R) % P).to_bytes(48, "little")


def fp_from_mont_bytes(b: bytes) -> int:
    v = int.from_bytes(b, "little")
    assert v < P, "Fp limb value not fully reduced"
    return (v * FP_RINV) % P


def fp_mont_limbs(a: int):
    v = (a * FP_R) % P
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]


def fr_to_mont_bytes(s: int) -> bytes:
    return ((s * FR_R) % R_ORDER).to_bytes(32, "little")


def fr_from_mont(b: bytes) -> int:
    """Scalar::into_bigint (src/scalar.rs:503-505): Montgomery blst_fr -> canonical integer."""
    return (int.from_bytes(b, "little") * FR_RINV) % R_ORDER


def fr_to_canon_bytes(s: int) -> bytes:
    return (s % R_ORDER).to_bytes(32, "little")


def _felt_bytes(F, a) -> bytes:
    return fp_to_mont_bytes(a) if F is F1 else fp_to_mont_bytes(a[0]) + fp_to_mont_bytes(a[1])


def _felt_from(F, b: bytes):
    return fp_from_mont_bytes(b) if F is F1 else (fp_from_mont_bytes(b[:48]), fp_from_mont_bytes(b[48:]))


def affine_to_bytes(F, p) -> bytes:
    """blst_p1_affine (96 B) / blst_p2_affine (192 B); infinity = all-zero."""
    n = 48 * (1 if F is F1 else 2)
    if p is INF:
        return bytes(2 * n)
    return _felt_bytes(F, p[0]) + _felt_bytes(F, p[1])


def affine_from_bytes(F, b: bytes):
    n = 48 * (1 if F is F1 else 2)
    assert len(b) == 2 * n
    if b == bytes(2 * n):
        return INF
    return (_felt_from(F, b[:n]), _felt_from(F, b[n:]))


def jac_from_bytes(F, b: bytes):
    """blst_p1 (144 B) / blst_p2 (288 B) -> affine point (canonical comparison form)."""
    n = 48 * (1 if F is F1 else 2)
    assert len(b) == 3 * n
    X, Y, Z = _felt_from(F, b[:n]), _felt_from(F, b[n:2 * n]), _felt_from(F, b[2 * n:])
    return jac_to_aff(F, (X, Y, Z))


def jac_to_bytes(F, p) -> bytes:
    """Affine point -> a Jacobian encoding with Z = 1 (or Z = 0 for infinity)."""
    if p is INF:
        return _felt_bytes(F, F.zero) * 3
    return _felt_bytes(F, p[0]) + _felt_bytes(F, p[1]) + _felt_bytes(F, F.one)


# ZCash / IETF uncompressed encoding (src/g1.rs:358-384): canonical byte form used for "bit-exact"
def g1_uncompressed(p) -> bytes:
    if p is INF:
        return bytes([0x40]) + bytes(95)
    return p[0].to_bytes(48, "big") + p[1].to_bytes(48, "big")


def g2_uncompressed(p) -> bytes:
    if p is INF:
        return bytes([0x40]) + bytes(191)
    (x0, x1), (y0, y1) = p
    return x1.to_bytes(48, "big") + x0.to_bytes(48, "big") + y1.to_bytes(48, "big") + y0.to_bytes(48, "big")


# ----------------------------------------------------------------------------------------------
# ZCash / IETF point encoding + validity for G1 (src/g1.rs:358-431): CanonicalSerialize / Valid / CanonicalDeserialize
# delegate to blstrs to_compressed / to_uncompressed / from_*_unchecked / is_on_curve / is_torsion_free [ext].
# Status codes of the batched restatement: 0 ok, 1 malformed encoding (flags, x >= p, no square root),
# 2 not on the curve, 3 not in the prime-order subgroup.
# ----------------------------------------------------------------------------------------------
DESER_OK, DESER_BAD_ENCODING, DESER_NOT_ON_CURVE, DESER_NOT_IN_SUBGROUP = 0, 1, 2, 3


def g1_compress(p) -> bytes:
    if p is INF:
        return bytes([0xC0]) + bytes(47)
    x, y = p
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80
    if y > (P - 1) // 2:
        b[0] |= 0x20
    return bytes(b)


def g1_in_subgroup(p) -> bool:
    return p is INF or scalar_mul(F1, p, R_ORDER) is INF


def g1_deserialize(b: bytes, compressed: bool, validate: bool):
    """Returns (point or INF or None, status)."""
    size = 48 if compressed else 96
    assert len(b) == size
    c_flag, i_flag, s_flag = b[0] >> 7, (b[0] >> 6) & 1, (b[0] >> 5) & 1
    if c_flag != (1 if compressed else 0):
        return None, DESER_BAD_ENCODING
    body = bytes([b[0] & 0x1F]) + b[1:]
    if i_flag:
        if any(body) or s_flag:
            return None, DESER_BAD_ENCODING
        return INF, DESER_OK
    x = int.from_bytes(body[:48], "big")
    if x >= P:
        return None, DESER_BAD_ENCODING
    rhs = (x * x * x + 4) % P
    if compressed:
        y = pow(rhs, (P + 1) // 4, P)
        if y * y % P != rhs:
            return None, DESER_BAD_ENCODING
        if (y > (P - 1) // 2) != bool(s_flag):
            y = P - y
    else:
        if s_flag:
            return None, DESER_BAD_ENCODING
        y = int.from_bytes(body[48:], "big")
        if y >= P:
            return None, DESER_BAD_ENCODING
    pt = (x, y)
    if validate:
        if y * y % P != rhs:
            return None, DESER_NOT_ON_CURVE
        if not g1_in_subgroup(pt):
            return None, DESER_NOT_IN_SUBGROUP
    return pt, DESER_OK


# ---- G2 (src/g2.rs:338-411): same format over Fp2, coordinates serialised c1 first; y^2 = x^3 + 4(1 + u)
def fp2_pow(a, e):
    r, b = (1, 0), a
    while e:
        if e & 1:
            r = F2.mul(r, b)
        b = F2.mul(b, b)
        e >>= 1
    return r


def fp2_sqrt(a):
    """Square root in Fp2 for p = 3 (mod 4) (Adj, Rodriguez-Henriquez, Alg. 9); None if a is not a square."""
    if F2.is_zero(a):
        return (0, 0)
    a1 = fp2_pow(a, (P - 3) // 4)
    alpha = F2.mul(a1, F2.mul(a1, a))
    x0 = F2.mul(a1, a)
    if alpha == (P - 1, 0):
        x = ((-x0[1]) % P, x0[0])
    else:
        b = fp2_pow(F2.add((1, 0), alpha), (P - 1) // 2)
        x = F2.mul(b, x0)
    return x if F2.eq(F2.mul(x, x), a) else None


def fp2_lex_largest(y) -> bool:
    """y > -y in the ZCash ordering: compare c1 first, then c0."""
    half = (P - 1) // 2
    return y[1] > half or (y[1] == 0 and y[0] > half)


def g2_compress(p) -> bytes:
    if p is INF:
        return bytes([0xC0]) + bytes(95)
    (x0, x1), y = p
    b = bytearray(x1.to_bytes(48, "big") + x0.to_bytes(48, "big"))
    b[0] |= 0x80
    if fp2_lex_largest(y):
        b[0] |= 0x20
    return bytes(b)


def g2_in_subgroup(p) -> bool:
    return p is INF or scalar_mul(F2, p, R_ORDER) is INF


def g2_deserialize(b: bytes, compressed: bool, validate: bool):
    size = 96 if compressed else 192
    assert len(b) == size
    c_flag, i_flag, s_flag = b[0] >> 7, (b[0] >> 6) & 1, (b[0] >> 5) & 1
    if c_flag != (1 if compressed else 0):
        return None, DESER_BAD_ENCODING
    body = bytes([b[0] & 0x1F]) + b[1:]
    if i_flag:
        if any(body) or s_flag:
            return None, DESER_BAD_ENCODING
        return INF, DESER_OK
    x1, x0 = int.from_bytes(body[:48], "big"), int.from_bytes(body[48:96], "big")
    if x0 >= P or x1 >= P:
        return None, DESER_BAD_ENCODING
    x = (x0, x1)
    rhs = F2.add(F2.mul(F2.mul(x, x), x), F2.b)
    if compressed:
        y = fp2_sqrt(rhs)
        if y is None:
            return None, DESER_BAD_ENCODING
        if fp2_lex_largest(y) != bool(s_flag):
            y = F2.neg(y)
    else:
        if s_flag:
            return None, DESER_BAD_ENCODING
        y1, y0 = int.from_bytes(body[96:144], "big"), int.from_bytes(body[144:192], "big")
        if y0 >= P or y1 >= P:
            return None, DESER_BAD_ENCODING
        y = (y0, y1)
    pt = (x, y)
    if validate:
        if not F2.eq(F2.mul(y, y), rhs):
            return None, DESER_NOT_ON_CURVE
        if not g2_in_subgroup(pt):
            return None, DESER_NOT_IN_SUBGROUP
    return pt, DESER_OK


# ----------------------------------------------------------------------------------------------
# deterministic input generators shared by tests / bench
# ----------------------------------------------------------------------------------------------
_M64 = 0xFFFFFFFFFFFFFFFF


def sm64(seed: int, ctr: int) -> int:
    """SplitMix64 output function in counter mode: word `ctr` of the stream keyed by `seed`."""
    z = (seed + (ctr + 1) * 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def gen_scalar(seed: int, i: int) -> int:
    """Uniform in [0, r): 255-bit rejection, up to 8 attempts (same stream as oracle/msm_oracle.c)."""
    v = 0
    for t in range(8):
        v = 0
        for j in range(4):
            v |= sm64(seed, (i * 8 + t) * 4 + j) << (64 * j)
        v &= (1 << 255) - 1
        if v < R_ORDER:
            return v
    return v - R_ORDER


def gen_dlog(seed: int, i: int) -> int:
    """Discrete log k_i of synthetic base i (P_i = k_i * G): 255 bits reduced mod r, never 0."""
    v = 0
    for j in range(4):
        v |= sm64(seed, i * 4 + j) << (64 * j)
    v &= (1 << 255) - 1
    if v >= R_ORDER:
        v -= R_ORDER
    return v or 1


def rand_scalars(n: int, seed: int):
    return [gen_scalar(seed, i) for i in range(n)]


# ----------------------------------------------------------------------------------------------
# self-check against the reference's embedded known-answer constants
# ----------------------------------------------------------------------------------------------
def selfcheck() -> None:
    x = X_PARAM
    # src/fp.rs:25-32
    ref_p = [0xB9FEFFFFFFFFAAAB, 0x1EABFFFEB153FFFF, 0x6730D2A0F6B0F624, 0x64774B84F38512BF, 0x4B1BA7B6434BACD7, 0x1A0111EA397FE69A]
    assert sum(l << (64 * i) for i, l in enumerate(ref_p)) == P
    assert P == (x - 1) ** 2 * (x ** 4 - x ** 2 + 1) // 3 + x
    # src/scalar.rs:476-481
    ref_r = [0xFFFFFFFF00000001, 0x53BDA402FFFE5BFE, 0x3339D80809A1D805, 0x73EDA753299D7D48]
    assert sum(l << (64 * i) for i, l in enumerate(ref_r)) == R_ORDER
    assert R_ORDER == x ** 4 - x ** 2 + 1
    # src/fp.rs:714-721 : Montgomery limbs of (p-1)/2
    kat = [0xA1FAFFFFFFFE5557, 0x995BFFF976A3FFFE, 0x03F41D24D174CEB4, 0xF6547998C1995DBD, 0x778A468F507A6034, 0x020559931F7F8103]
    assert fp_mont_limbs((P - 1) // 2) == kat
    # src/g1.rs:42-51
    assert H1 == (x - 1) ** 2 // 3 == 76329603384216526031706109802092473003
    assert (0x396C8C005555E156 << 64) | 0x8C00AAAB0000AAAB == H1
    cof_inv_limbs = [288839107172787499, 1152722415086798946, 2612889808468387987, 5124657601728438008]
    cof_inv = fr_from_mont(b"".join(l.to_bytes(8, "little") for l in cof_inv_limbs))
    assert cof_inv == 52435875175126190458656871551744051925719901746859129887267498875565241663483
    assert (cof_inv * H1) % R_ORDER == 1
    # src/g2.rs:45-63
    h2_limbs = [0xCF1C38E31C7238E5, 0x1616EC6E786F0C70, 0x21537E293A6691AE, 0xA628F1CB4D9E82EF,
                0xA68A205B2E5A7DDF, 0xCD91DE4547085ABA, 0x091D50792876A202, 0x05D543A95414E7F1]
    h2 = sum(l << (64 * i) for i, l in enumerate(h2_limbs))
    assert h2 == 305502333931268344200999753193121504214466019254188142667664032982267604182971884026507427359259977847832272839041616661285803823378372096355777062779109
    h2_inv_limbs = [6746407649509787816, 1304054119431494378, 2461312685643913071, 5956596749362435284]
    h2_inv = fr_from_mont(b"".join(l.to_bytes(8, "little") for l in h2_inv_limbs))
    assert h2_inv == 26652489039290660355457965112010883481355318854675681319708643586776743290055
    assert (h2_inv * h2) % R_ORDER == 1
    # generators on curve, of order r, and encode/decode round trips
    assert on_curve(F1, G1_GEN) and on_curve(F2, G2_GEN)
    assert scalar_mul(F1, G1_GEN, R_ORDER) is INF and scalar_mul(F2, G2_GEN, R_ORDER) is INF
    assert affine_from_bytes(F1, affine_to_bytes(F1, G1_GEN)) == G1_GEN
    assert affine_from_bytes(F2, affine_to_bytes(F2, G2_GEN)) == G2_GEN
    assert fp_mont_limbs(1) == [0x760900000002FFFD, 0xEBF4000BC40C0002, 0x5F48985753C758BA, 0x77CE585370525745, 0x5C071A97A256EC6D, 0x15F65EC3FA80E493]
    two_g = scalar_mul(F1, G1_GEN, 2)
    assert two_g[0] == 0x0572CBEA904D67468808C8EB50A9450C9721DB309128012543902D0AC358A62AE28F75BB8F1C7C42C39A8C5529BF0F4E
    assert aff_add(F1, G1_GEN, G1_GEN) == two_g


if __name__ == "__main__":
    selfcheck()
    print("oracle/bls12_381.py selfcheck OK")
