#!/usr/bin/env python3
"""Time-boxed randomized parity soak on one MI355X: MSM (G1/G2) and multi-pairing through the C ABI against the C
oracles, over random sizes, window sizes (7..22), scalar formats/distributions, repeated / opposite / infinity bases, resident
plain and precomputed-table base sets, and (single-threaded runs, through the test build's hooks) calls split into several passes and
pairings with forced pairs-per-accumulator / line-buffer batch sizes.
Prints a progress line every ~20 s and a final JSON summary; exits non-zero on the first mismatch.
    python tools/soak.py [seconds=300] [seed=1] [threads=1]
With threads > 1 the same loop runs from several host threads on ONE context (two lanes + exclusive entry points); the batch
case, which changes the resident base set, is then left out.
"""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
try:
    import torch   # first: its bundled HIP runtime must be the one the process loads (load order)
    if not torch.cuda.is_available():
        torch = None
except Exception:
    torch = None
import __graft_entry__ as ge
from oracle import coracle as co, bls12_381 as o
pkg = ge.load_package()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nthreads = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rnd = random.Random(seed)
ncpu = min(16, len(os.sched_getaffinity(0)))
POOL = {"g1": 200000, "g2": 20000}   # G1 calls above 2^16 / 2^17 points cross PCIe in several chunks (host slices)
pools = {g: co.gen_bases(g, 900 + seed + i, POOL[g], ncpu) for i, g in enumerate(("g1", "g2"))}
AFF = {"g1": 96, "g2": 192}


def neg_point(g, blob):
    F = o.F1 if g == "g1" else o.F2
    return o.affine_to_bytes(F, o.aff_neg(F, o.affine_from_bytes(F, blob)))


stats = {"msm_g1": 0, "msm_g2": 0, "pairing": 0, "points": 0, "pairs": 0}
t0 = last = time.time()
import threading
stats_mu = threading.Lock()
failed = []


def loop(ctx, rnd, tid):
    global last
    while time.time() - t0 < budget and not failed:
        r = rnd.random()
        if r < 0.03 and nthreads == 1:   # precomputed 2^(c j) P tables: resident set, several calls over prefixes of it
            g = rnd.choice(["g1", "g1", "g2"])
            aff, lim = AFF[g], POOL[g] // 3
            n = rnd.randrange(1, lim)
            start = rnd.randrange(0, POOL[g] - n + 1)
            bases = bytearray(pools[g][aff * start:aff * (start + n)])
            for k in range(min(n, 8)):
                if rnd.random() < 0.5: bases[aff * rnd.randrange(n):][:aff] = bytes(aff)
            j = rnd.randrange(n)
            bases[aff * j:aff * (j + 1)] = bytes(aff)
            bases = bytes(bases)
            ctx.set_bases_precomputed(g, bases, n, rnd.choice([0, 0, 8, 9, 11, 13, 14, 16]))
            for _ in range(rnd.randrange(1, 4)):
                m = rnd.choice([n, n, rnd.randrange(1, n + 1)])
                kind = rnd.choice(["uniform", "bits", "small", "edge"])
                if kind == "uniform": sc = [rnd.randrange(o.R_ORDER) for _ in range(m)]
                elif kind == "bits": sc = [rnd.randrange(2) for _ in range(m)]
                elif kind == "small": sc = [rnd.randrange(1 << rnd.choice([8, 40, 128])) for _ in range(m)]
                else: sc = [rnd.choice([0, 1, 2, o.R_ORDER - 1, o.R_ORDER - 2, 1 << 254]) for _ in range(m)]
                canon = b"".join(o.fr_to_canon_bytes(x) for x in sc)
                got = ctx.msm(g, None, canon, m, pkg.SCALAR_CANONICAL)
                if co.to_affine(g, got) != co.to_affine(g, co.msm(g, bases, canon, m, 0, ncpu)):
                    print("PRECOMPUTED MISMATCH", seed, g, n, m, kind); failed.append(1); return
                stats["precomputed"] = stats.get("precomputed", 0) + 1; stats["points"] += m
            ctx.set_bases(g, bases[:aff], 1)   # drop the tables (the W x allocation is given back: DevBuf::ensure_fit)
        elif r < 0.045 and nthreads == 1:   # round 5: a resident set through Valid::check on the GPU, then MSMs with the sign fold (c = 15 / 17 lose a window)
            g = rnd.choice(["g1", "g1", "g2"])
            aff, lim = AFF[g], POOL[g] // 2
            n = rnd.randrange(1, lim)
            start = rnd.randrange(0, POOL[g] - n + 1)
            bases = bytearray(pools[g][aff * start:aff * (start + n)])
            if rnd.random() < 0.5: bases[aff * rnd.randrange(n):][:aff] = bytes(aff)   # infinity passes the check
            bases = bytes(bases)
            ctx.set_bases(g, bases, n)
            if ctx.validate_bases(g) != 0:
                print("VALIDATE MISMATCH: subgroup points rejected", seed, g, n); failed.append(1); return
            for cb in (15, 17, rnd.choice([0, 13, 16])):
                m = rnd.choice([n, rnd.randrange(1, n + 1)])
                sc = [rnd.choice([rnd.randrange(o.R_ORDER), o.R_ORDER - 1 - rnd.randrange(1000), (o.R_ORDER + 1) // 2 + rnd.randrange(1000)]) for _ in range(m)]
                canon = b"".join(o.fr_to_canon_bytes(x) for x in sc)
                ctx.set_window_bits(cb)
                try:
                    got = ctx.msm(g, None, canon, m, pkg.SCALAR_CANONICAL)
                    folded = ctx.profile()["num_windows"] == -(-255 // cb) if cb in (15, 17) else True
                finally:
                    ctx.set_window_bits(0)
                if not folded or co.to_affine(g, got) != co.to_affine(g, co.msm(g, bases, canon, m, 0, ncpu)):
                    print("VALIDATED MISMATCH", seed, g, n, m, cb, folded); failed.append(1); return
                stats["validated"] = stats.get("validated", 0) + 1; stats["points"] += m
            ctx.set_bases(g, bases[:aff], 1)
        elif r < 0.06 and nthreads == 1:   # batch entry point over a resident base set (two MSMs in flight on the context's lanes)
            g = rnd.choice(["g1", "g1", "g2"])
            aff, lim = AFF[g], POOL[g]
            n = rnd.randrange(1, lim)
            start = rnd.randrange(0, lim - n + 1)
            bases = pools[g][aff * start:aff * (start + n)]
            ctx.set_bases(g, bases, n)
            k = rnd.randrange(1, 5)
            vecs = [b"".join(o.fr_to_canon_bytes(rnd.randrange(o.R_ORDER)) for _ in range(n)) for _ in range(k)]
            got = ctx.msm_batch(g, vecs, n, pkg.SCALAR_CANONICAL)
            for v, x in zip(vecs, got):
                if co.to_affine(g, x) != co.to_affine(g, co.msm(g, bases, v, n, 0, ncpu)):
                    print("BATCH MISMATCH", seed, g, n, k); failed.append(1); return
            stats["batch"] = stats.get("batch", 0) + 1; stats["points"] += n * k
        elif r < 0.09 and nthreads == 1 and torch is not None:   # exchange entry points: window sums left in device memory, folded on the host
            g = rnd.choice(["g1", "g1", "g2"])
            aff, size = AFF[g], (144 if g == "g1" else 288)
            n = rnd.randrange(1, POOL[g] // 2)
            ranks = rnd.randrange(1, 4)
            cuts = sorted([0, n] + [rnd.randrange(0, n + 1) for _ in range(ranks - 1)])
            start = rnd.randrange(0, POOL[g] - n + 1)
            bases = pools[g][aff * start:aff * (start + n)]
            canon = b"".join(o.fr_to_canon_bytes(rnd.randrange(o.R_ORDER)) for _ in range(n))
            d = torch.frombuffer(bytearray(canon), dtype=torch.uint8).cuda()
            win = torch.zeros(pkg.MAX_WINDOWS * size, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            c = rnd.choice([9, 11, 13, 16])   # ragged shards: one window size for all of them
            ctx.set_window_bits(c)
            gathered, info = b"", None
            try:
                for k in range(ranks):
                    lo, hi = cuts[k], cuts[k + 1]
                    if hi == lo:
                        continue
                    ctx.set_bases(g, bases[aff * lo:aff * hi], hi - lo)
                    info = ctx.msm_device_windows(g, d.data_ptr() + 32 * lo, hi - lo, pkg.SCALAR_CANONICAL, win.data_ptr())
                    gathered += win.cpu().numpy().tobytes()[:info[1] * size]
            finally:
                ctx.set_window_bits(0)
            got = pkg.fold_windows(g, gathered, len(gathered) // (info[1] * size), info[1], *info)
            if co.to_affine(g, got) != co.to_affine(g, co.msm(g, bases, canon, n, 0, ncpu)):
                print("WINDOWS MISMATCH", seed, g, n, ranks, c); failed.append(1); return
            stats["windows"] = stats.get("windows", 0) + 1; stats["points"] += n
        elif r < 0.15:
            n = rnd.choice([1, 2, 9, 10, 11, 63, 64, 65, rnd.randrange(1, 3000)])
            i1 = [rnd.randrange(POOL["g1"]) for _ in range(n)]
            i2 = [rnd.randrange(POOL["g2"]) for _ in range(n)]
            g1 = bytearray(b"".join(pools["g1"][96 * i:96 * i + 96] for i in i1))
            g2 = bytearray(b"".join(pools["g2"][192 * i:192 * i + 192] for i in i2))
            for k in range(n):
                if rnd.random() < 0.03: g1[96 * k:96 * k + 96] = bytes(96)
                if rnd.random() < 0.03: g2[192 * k:192 * k + 192] = bytes(192)
            forced = nthreads == 1 and rnd.random() < 0.5   # pairs per accumulator / line-buffer batch as large inputs get them
            if forced: ctx.test_set_pairing(share=rnd.randrange(1, 9), batch=rnd.choice([0, 0, 100, 777]))
            try:
                got = ctx.multi_pairing(bytes(g1), bytes(g2))
            finally:
                if forced: ctx.test_set_pairing()
            want = co.multi_pairing(bytes(g1), bytes(g2), ncpu)
            if got != want:
                print("PAIRING MISMATCH", seed, n); failed.append(1); return
            stats["pairing"] += 1; stats["pairs"] += n
        elif r < 0.30 and nthreads == 1:   # round 6: SRS loading as one call (decode + check on the GPU) and normalize feeding the resident set, then MSMs over it
            g = rnd.choice(["g1", "g1", "g2"])
            aff, lim = AFF[g], POOL[g] // 4
            n = rnd.choice([1, 2, 255, 256, 257, rnd.randrange(1, lim), rnd.randrange(1, lim)])
            start = rnd.randrange(0, POOL[g] - n + 1)
            bases = bytearray(pools[g][aff * start:aff * (start + n)])
            for k in range(min(n, 6)):
                if rnd.random() < 0.5: bases[aff * rnd.randrange(n):][:aff] = bytes(aff)
            bases = bytes(bases)
            compressed = rnd.random() < 0.7
            enc = ctx.serialize_batch(g, bases, compressed)
            dec, st = ctx.deserialize_batch(g, enc, compressed, True)
            if dec != bases or st != bytes(n):
                print("CODEC MISMATCH", seed, g, n, compressed); failed.append(1); return
            if rnd.random() < 0.5:
                if ctx.set_bases_from_compressed(g, enc, n, compressed, rnd.random() < 0.7) != 0:
                    print("SRS LOAD REJECTED VALID POINTS", seed, g, n); failed.append(1); return
            else:   # Jacobian with Z = 1 or a scaled Z for a few points (the C oracle normalises them back)
                one = bytes.fromhex("fdff02000000097602000cc40b00f4ebba58c7535798485f455752705358ce776dec56a2971a075c93e480fac35ef615") + (bytes(48) if g == "g2" else b"")
                jac = b"".join((bases[aff * i:aff * (i + 1)] + one) if any(bases[aff * i:aff * (i + 1)]) else bytes(aff + len(one)) for i in range(n))
                if co.normalize_batch(g, jac, ncpu) != bases or ctx.normalize_batch(g, jac) != bases:
                    print("NORMALIZE MISMATCH", seed, g, n); failed.append(1); return
                ctx.set_bases_from_jacobian(g, jac, n)
            for rep in range(2):
                m = rnd.choice([n, rnd.randrange(1, n + 1)])
                sc = [rnd.randrange(o.R_ORDER) for _ in range(m)]
                canon = b"".join(o.fr_to_canon_bytes(x) for x in sc)
                ctx.set_pipeline(rnd.choice([None, [1, 3], [1, 1, 1]]))
                got = ctx.msm(g, None, canon, m, pkg.SCALAR_CANONICAL)
                ctx.set_pipeline(None)
                if co.to_affine(g, got) != co.to_affine(g, co.msm(g, bases[:aff * m], canon, m, 0, ncpu)):
                    print("MSM OVER LOADED SRS MISMATCH", seed, g, n, m); failed.append(1); return
                stats["msm_" + g] += 1; stats["points"] += m
            stats["srs_loads"] = stats.get("srs_loads", 0) + 1
        else:
            g = "g2" if r < 0.4 else "g1"
            aff, lim = AFF[g], POOL[g]
            n = rnd.choice([1, 2, 3, 64, 65, 1000, rnd.randrange(1, 2000), rnd.randrange(1, lim), rnd.randrange(1, lim)])
            idx = [rnd.randrange(lim) for _ in range(n)]
            if rnd.random() < 0.4:
                idx = [idx[rnd.randrange(max(1, n // rnd.choice([2, 4, 50])))] for _ in range(n)]
            bases = bytearray(b"".join(pools[g][aff * i:aff * (i + 1)] for i in idx))
            for k in range(min(n, 50)):
                q = rnd.random()
                j = rnd.randrange(n)
                if q < 0.3: bases[aff * j:aff * (j + 1)] = bytes(aff)
                elif q < 0.6: bases[aff * j:aff * (j + 1)] = neg_point(g, bytes(bases[aff * rnd.randrange(n):][:aff]) or bytes(aff)) if any(bases[aff * j:aff * (j + 1)]) else bytes(aff)
            kind = rnd.choice(["uniform", "uniform", "bits", "small", "equal", "edge", "fewvalues"])
            if kind == "uniform": sc = [rnd.randrange(o.R_ORDER) for _ in range(n)]
            elif kind == "bits": sc = [rnd.randrange(2) for _ in range(n)]
            elif kind == "small": sc = [rnd.randrange(1 << rnd.choice([8, 16, 40, 64, 128])) for _ in range(n)]
            elif kind == "equal": sc = [rnd.randrange(o.R_ORDER)] * n
            elif kind == "fewvalues":
                vals = [rnd.randrange(o.R_ORDER) for _ in range(rnd.choice([2, 3, 8]))]
                sc = [rnd.choice(vals) for _ in range(n)]
            else: sc = [rnd.choice([0, 1, 2, o.R_ORDER - 1, o.R_ORDER - 2, (o.R_ORDER - 1) // 2, 1 << 254]) for _ in range(n)]
            canon = b"".join(o.fr_to_canon_bytes(x) for x in sc)
            fmt = rnd.choice([pkg.SCALAR_CANONICAL, pkg.SCALAR_MONTGOMERY])
            data = canon if fmt == pkg.SCALAR_CANONICAL else co.fr_to_mont(canon)
            c = rnd.choice([0, 0, 0, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, rnd.randrange(17, 23)]) if nthreads == 1 else 0   # the setting is per context
            part = rnd.choice([0] * 9 + [rnd.randrange(200, 5000)]) if nthreads == 1 else 0   # passes per call (test hook)
            pipe = rnd.choice([None, None, [1], [1, 3], [1, 1], [2, 1, 1], [1, 1, 1, 1], [1, 7], [5, 1]]) if nthreads == 1 else None   # round 6: window groups
            if nthreads == 1:
                ctx.set_window_bits(c)
                ctx.test_set_max_part(part)
                ctx.set_pipeline(pipe)
            try:
                got = ctx.msm(g, bytes(bases), data, n, fmt)
            finally:
                if nthreads == 1:
                    ctx.set_window_bits(0)
                    ctx.test_set_max_part(0)
                    ctx.set_pipeline(None)
            want = co.msm(g, bytes(bases), canon, n, 0, ncpu)
            if co.to_affine(g, got) != co.to_affine(g, want):
                print("MSM MISMATCH", seed, g, n, kind, c, fmt); failed.append(1); return
            stats["msm_" + g] += 1; stats["points"] += n
        if tid == 0 and time.time() - last > 20:
            last = time.time()
            print("soak", round(last - t0), "s", stats, flush=True)


with pkg.Context([0], test_hooks=True) as ctx:   # the test build: the same sources plus the mi_test_* hooks
    # round 5: the base-set cache ON for every host-bases call of 4096 points or more.  The loop's base vectors are fresh `bytes` objects:
    # the allocator hands the same address (and length) out again with other content — exactly what the fingerprint has to catch
    ctx.set_base_cache(2)
    ths = [threading.Thread(target=loop, args=(ctx, random.Random(seed * 1000 + t), t)) for t in range(nthreads)]
    for t in ths: t.start()
    for t in ths: t.join()
    stats["base_cache"] = ctx.base_cache_stats()
if failed:
    sys.exit(1)
stats.update({"seconds": round(time.time() - t0, 1), "seed": seed, "threads": nthreads, "mismatches": 0})
print(json.dumps(stats))
