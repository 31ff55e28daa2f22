"""Host-side mirror of the reference's operator surface for the MSM path.

Reference (Rust, /root/reference):  impl VariableBaseMSM for G1Projective { fn msm(bases: &[G1Affine],
scalars: &[Scalar]) -> Result<Self, usize> }  (src/g1.rs:602-632; G2: src/g2.rs:582-612), with
`ScalarMul::MulBase = G1Affine` (src/g1.rs:593-600).  Same names and argument meaning here; the Result<_, usize>
error convention is kept: arkworks' generic entry returns Err(min(len)) on a length mismatch and the
reference's GPU impl returns Err(0) for any device failure (src/g1.rs:628-630) — mirrored by MsmErr.
"""
from __future__ import annotations

from .binding import Context, MsmError, SCALAR_CANONICAL, SCALAR_MONTGOMERY

_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
        # the trait call carries no handle: keep the device form of the last two base vectors (mi_msm_set_base_cache, include/arkblst_amd.h:
        # keyed by a fingerprint of every byte of the vector, so msm() stays a function of its arguments); ARKBLST_AMD_BASE_CACHE=0 in the
        # environment switches it off
        _default_ctx.set_base_cache(2)
    return _default_ctx


class MsmErr(Exception):
    """Err(usize) of the reference's Result: .value = min(len) on length mismatch, 0 on a GPU failure."""

    def __init__(self, value: int, detail: str = ""):
        self.value = value
        super().__init__(f"Err({value}) {detail}")


class _Projective:
    GROUP = ""
    AFFINE_BYTES = 0

    @classmethod
    def msm(cls, bases: bytes, scalars: bytes, *, scalar_fmt: int = SCALAR_MONTGOMERY, ctx: Context | None = None) -> bytes:
        """bases: packed blst affine points; scalars: packed 32-byte `Scalar`s (Montgomery blst_fr, the in-memory
        form of the reference's `Scalar`, by default).  Returns the projective result (blst_p1 / blst_p2 bytes)."""
        nb, ns = len(bases) // cls.AFFINE_BYTES, len(scalars) // 32
        if nb != ns:
            raise MsmErr(min(nb, ns), "bases and scalars differ in length")
        try:
            return (ctx or default_context()).msm(cls.GROUP, bases, scalars, nb, scalar_fmt)
        except MsmError as e:
            raise MsmErr(0, str(e)) from e

    @classmethod
    def msm_bigint(cls, bases: bytes, bigints: bytes, *, ctx: Context | None = None) -> bytes:
        """arkworks' msm_bigint: scalars already canonical BigInteger256 (what src/g1.rs:624-627 builds)."""
        return cls.msm(bases, bigints, scalar_fmt=SCALAR_CANONICAL, ctx=ctx)


class G1Projective(_Projective):
    GROUP = "g1"
    AFFINE_BYTES = 96


class G2Projective(_Projective):
    GROUP = "g2"
    AFFINE_BYTES = 192
