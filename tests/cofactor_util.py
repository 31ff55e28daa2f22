"""Shared by tests/test_oracle.py (CPU) and tests/test_gpu_msm.py (GPU): curve points OUTSIDE the prime-order subgroup and the
reference-held cofactors (/root/reference/src/g1.rs:42, src/g2.rs:45-54) that must carry them INTO it.

The reference holds no MSM vector, but it does hold h1 and h2.  For a point P of E(Fp) (resp. E'(Fp2)) that is not in the
r-torsion, h * P lies in it exactly when h is the cofactor and the group law is right: r * (h * P) = infinity while r * P is not.
That ties the addition / doubling formulas and the scalar path of every implementation to a reference-held constant, not only to
each other (DESIGN_HISTORY.md §3: "group law pinned by cofactor clearing")."""

H1_LIMBS = [0x8C00AAAB0000AAAB, 0x396C8C005555E156]                       # src/g1.rs:42
H2_LIMBS = [0xCF1C38E31C7238E5, 0x1616EC6E786F0C70, 0x21537E293A6691AE, 0xA628F1CB4D9E82EF,
            0xA68A205B2E5A7DDF, 0xCD91DE4547085ABA, 0x091D50792876A202, 0x05D543A95414E7F1]   # src/g2.rs:45-54
H1 = sum(l << (64 * i) for i, l in enumerate(H1_LIMBS))
H2 = sum(l << (64 * i) for i, l in enumerate(H2_LIMBS))
SPLIT_BITS = 250   # h2 (507 bits) = a0 + a1 2^250 + a2 2^500 with every piece below r (255 bits)


def off_subgroup_points(o, group, count=3):
    """the first `count` points with small x on the curve of `group` that are NOT in the r-torsion (almost all points: the
    cofactors are 126 and 507 bits)"""
    F = o.F1 if group == "g1" else o.F2
    out, k = [], 0
    while len(out) < count:
        k += 1
        x = k if group == "g1" else (k, 1)
        rhs = F.add(F.mul(F.mul(x, x), x), F.b)
        if group == "g1":
            y = pow(rhs, (o.P + 1) // 4, o.P)
            if (y * y - rhs) % o.P:
                continue
        else:
            y = o.fp2_sqrt(rhs)
            if y is None:
                continue
        pt = (x, y)
        assert o.on_curve(F, pt)
        if o.scalar_mul(F, pt, o.R_ORDER) is not o.INF:
            out.append(pt)
    return out


def h2_pieces():
    m = (1 << SPLIT_BITS) - 1
    return [H2 & m, (H2 >> SPLIT_BITS) & m, H2 >> (2 * SPLIT_BITS)]
