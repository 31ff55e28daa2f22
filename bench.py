#!/usr/bin/env python3
"""bench.py — G1 MSM points/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n 20] [--group g1]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
              bench.py --gpus N --steps K --warmup W

A "step" is one full MSM over one batch of 2^log_n synthetic (base, scalar) pairs PER GPU through the C ABI
(mi_msm_g1_device): digit extraction, bucket sort, bucket accumulation, bucket reduction, host fold — nothing is
cached between steps.  Inputs are resident in HBM when the timed region starts (bases as the resident SRS,
scalars in a device buffer).  With N ranks the base set is N * 2^log_n points sharded contiguously (weak scaling,
no data-path collective); each step ends with the RCCL all-gather of the N 144-byte partial sums and the
deterministic fold on every rank.  The result of the last step is checked bit-exact (canonical affine bytes)
against the closed form (sum s_i k_i) G.

The oracle (oracle/) is used ONLY to generate the synthetic inputs, as the checker, and as the timed CPU baseline
(`cpu_baseline`, kind "port": the reference's blst path cannot be built in this image).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED_B, SEED_S = 0xA55E7 + 2, 0x5CA1A5 + 2
ALG_BYTES_PER_POINT = {"g1": 128, "g2": 224}   # base + scalar, each read once
MADS_PER_POINT = {"g1": 48_000, "g2": 144_000} # SURVEY.md §8(d) canonical integer-op model
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: 8 TB/s spec
MAD_PEAK_TLOPS = 39.3                          # 1024 SIMD x 64 lanes x 2.4 GHz / 4 cyc (v_mad_u64_u32 is half rate:
                                               # tools/ubench_valu.hip measured 33.4 T/s incl. loop overhead)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=20, help="log2 of points PER GPU")
    ap.add_argument("--group", default="g1", choices=["g1", "g2"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--dist", default="uniform", choices=["uniform", "zero_one", "small64", "all_equal", "all_ones", "r1cs_mix"],
                    help="scalar distribution (secondary robustness figures; the headline is uniform)")
    ap.add_argument("--concurrency", type=int, default=1,
                    help="host threads issuing MSM calls concurrently (each with its own context; the trait method is\n"
                         "re-entrant, SURVEY 8b).  1 = blocking calls back to back (headline).")
    ap.add_argument("--no-secondary", action="store_true", help="skip the two-host-thread secondary figure (profiling runs)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-device rehearses the N>1 path on a single-GPU box")
    ap.add_argument("--share-device", action="store_true", help="all ranks use GPU 0 (rehearsal only)")
    args = ap.parse_args()

    import torch  # plumbing only: device buffers, synchronize, torch.distributed (RCCL)
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the MSM path")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    on_gpu = args.backend == "nccl"   # collectives on device tensors (RCCL) or on host tensors (gloo rehearsal)
    if world > 1:
        if on_gpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    import __graft_entry__ as ge
    from oracle import coracle as co   # input generation, checker, cpu_baseline only

    pkg = ge.load_package()
    g = args.group
    n = 1 << args.log_n
    ncpu = _host_threads()
    aff = 96 if g == "g1" else 192

    # ---- synthetic inputs (seeded): rank r owns global indices [r*n, (r+1)*n)
    t0 = time.time()
    seed_b, seed_s = SEED_B + 1000 * rank, SEED_S + 1000 * rank
    bases = co.gen_bases(g, seed_b, n, ncpu)
    scalars = co.gen_scalars(seed_s, n)
    if args.dist != "uniform":
        import numpy as np
        a = np.frombuffer(scalars, dtype=np.uint8).reshape(n, 32).copy()
        if args.dist == "zero_one":      # R1CS-like witness: bits
            a[:, 1:] = 0
            a[:, 0] &= 1
        elif args.dist == "small64":     # 64-bit values
            a[:, 8:] = 0
        elif args.dist == "r1cs_mix":    # witness-like: half zeros, a quarter ones, a quarter full-size values
            sel = a[:, 0] & 3
            a[sel < 2] = 0
            a[sel == 2] = 0
            a[sel == 2, 0] = 1
        elif args.dist == "all_ones":    # every scalar = 1: the plain sum of the bases, one bucket of N entries
            a[:] = 0
            a[:, 0] = 1
        else:                            # every scalar identical: all points in one bucket per window
            a[:] = a[0]
        scalars = a.tobytes()
    gen_s = time.time() - t0

    ctx = pkg.Context([local_rank])
    if args.window_bits:
        ctx.set_window_bits(args.window_bits)
    ctx.set_bases(g, bases, n)
    d_scalars = torch.frombuffer(bytearray(scalars), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()

    jac_bytes = 144 if g == "g1" else 288
    cdev = "cuda" if on_gpu else "cpu"
    gather = torch.empty(world * jac_bytes, dtype=torch.uint8, device=cdev) if world > 1 else None
    mine_dev = torch.empty(jac_bytes, dtype=torch.uint8, device=cdev) if world > 1 else None
    # staging buffers for the exchange: pinned on a GPU box (no pageable-copy synchronisation per step)
    mine_host = torch.empty(jac_bytes, dtype=torch.uint8) if world > 1 else None
    gather_host = torch.empty(world * jac_bytes, dtype=torch.uint8) if world > 1 else None
    if world > 1 and on_gpu:
        mine_host, gather_host = mine_host.pin_memory(), gather_host.pin_memory()

    def step() -> bytes:
        part = ctx.msm_device(g, d_scalars.data_ptr(), n, pkg.SCALAR_CANONICAL)
        if world == 1:
            return part
        mine_host.numpy()[:] = memoryview(part)
        mine_dev.copy_(mine_host, non_blocking=True)
        dist.all_gather_into_tensor(gather, mine_dev)       # RCCL over xGMI: N x 144 B (latency-bound)
        gather_host.copy_(gather, non_blocking=True)        # one D2H copy of the N partials
        if on_gpu:
            torch.cuda.current_stream().synchronize()
        allp = gather_host.numpy().tobytes()
        # all-reduce under the curve group law: fold in rank order on every rank (identical result everywhere)
        return (pkg.g1_sum if g == "g1" else pkg.g2_sum)([allp[k * jac_bytes:(k + 1) * jac_bytes] for k in range(world)])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    prof_acc = []
    if args.concurrency > 1 and world == 1:
        # K steps issued from `concurrency` host threads, each with its own context and stream on the same GPU
        import threading
        ctxs = [ctx] + [pkg.Context([local_rank]) for _ in range(args.concurrency - 1)]
        for c in ctxs[1:]:
            c.set_bases(g, bases, n)
            c.msm_device(g, d_scalars.data_ptr(), n, pkg.SCALAR_CANONICAL)
        results = [b""] * args.steps
        def worker(t):
            for k in range(t, args.steps, args.concurrency):
                results[k] = ctxs[t].msm_device(g, d_scalars.data_ptr(), n, pkg.SCALAR_CANONICAL)
        fence()
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(t,)) for t in range(args.concurrency)]
        for x in th: x.start()
        for x in th: x.join()
        fence()
        elapsed = time.perf_counter() - t0
        result = results[-1]
        # equal as curve points (the Jacobian representative depends on the order entries reached their bucket)
        assert len({co.to_affine(g, r) for r in results}) == 1
        prof_acc.append(ctx.profile())
        for c in ctxs[1:]:
            c.close()
    else:
        fence()
        t0 = time.perf_counter()
        result = b""
        for _ in range(args.steps):
            result = step()
            prof_acc.append(ctx.profile())
        fence()
        elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- secondary figure (N = 1 only): the same K steps issued by TWO host threads on the same context (the trait
    # method is re-entrant and arkworks calls it from rayon workers): a context has two lanes, so one call's
    # sort / reduce / host fold overlap the other's accumulate.  Not the headline value.
    two_thread = None
    if world == 1 and args.concurrency == 1 and not args.no_secondary:
        import threading
        res2 = [b""] * args.steps
        def worker(t):
            for k in range(t, args.steps, 2):
                res2[k] = ctx.msm_device(g, d_scalars.data_ptr(), n, pkg.SCALAR_CANONICAL)   # same context: its two lanes
        worker(1)   # first use of the second lane allocates its scratch
        fence()
        t1 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
        for x in th: x.start()
        for x in th: x.join()
        fence()
        e2 = time.perf_counter() - t1
        ok2 = len({co.to_affine(g, r) for r in res2 + [result]}) == 1
        two_thread = {"value": n * args.steps / e2, "unit": "points/s", "ms_per_step": e2 / args.steps * 1e3, "same_result": ok2,
                      "note": "the same K steps issued by two host threads on ONE context (two lanes, shared resident bases)"}

    # ---- secondary figure (N = 1 only): end-to-end call shapes (SURVEY 8(d) "timing scope"): scalars from host memory per call
    # with resident bases, and the reference driver's shape — bases AND scalars uploaded on every call (src/gpu.rs:149-150)
    call_shapes = None
    if world == 1 and args.concurrency == 1 and not args.no_secondary:
        def best_of(fn, reps=3):
            fn()
            b = 1e30
            for _ in range(reps):
                t1 = time.perf_counter(); r = fn(); b = min(b, time.perf_counter() - t1)
            assert co.to_affine(g, r) == co.to_affine(g, result)
            return b * 1e3
        call_shapes = {"resident_bases_host_scalars_ms": best_of(lambda: ctx.msm(g, None, scalars, n, pkg.SCALAR_CANONICAL)),
                       "host_bases_host_scalars_ms": best_of(lambda: ctx.msm(g, bases, scalars, n, pkg.SCALAR_CANONICAL)),
                       "note": "per call incl. H2D of the scalars (and bases) from pageable host memory; the headline keeps both in HBM"}

    # ---- secondary figure (N = 1 only): the pairing row (BASELINE config #5): 2^16 G1 x G2 pairs through
    # mi_multi_pairing; parity = prod e(P_i, Q_i) e(-P_i, Q_i) == 1 at full size (tools/bench_pairing.py has the oracle check)
    pairing = None
    if world == 1 and args.concurrency == 1 and not args.no_secondary:
        try:
            from oracle import pairing as pr_oracle, bls12_381 as o
            half = 1 << 15
            p1 = co.gen_bases("g1", seed_b + 101, half, ncpu)
            q2 = co.gen_bases("g2", seed_b + 102, half, ncpu)
            neg = bytearray(p1)
            for i in range(half):   # -P: y -> p - y (Montgomery form, y != 0)
                y = int.from_bytes(p1[96 * i + 48:96 * i + 96], "little")
                neg[96 * i + 48:96 * i + 96] = (o.P - y).to_bytes(48, "little")
            P_all, Q_all = p1 + bytes(neg), q2 + q2
            ctx.multi_pairing(P_all[:96 * 64], Q_all[:192 * 64])
            best = 1e30
            for _ in range(3):
                t1 = time.perf_counter()
                gt = ctx.multi_pairing(P_all, Q_all)
                best = min(best, time.perf_counter() - t1)
            pp = ctx.profile()
            pairing = {"metric": "pairs/s, batched Miller loop + final exponentiation (host buffers in, Gt out)", "value": 2 * half / best,
                       "n_pairs": 2 * half, "ms": best * 1e3, "miller_kernels_ms": pp["accumulate_ms"], "fp12_tree_ms": pp["reduce_ms"],
                       "h2d_ms": pp["h2d_ms"], "host_tail_ms": pp["host_fold_ms"],
                       "product_cancels_to_one": gt == pr_oracle.fp12_to_bytes(pr_oracle.FP12_ONE)}
        except Exception as e:   # never let a secondary figure break the headline line
            pairing = {"error": repr(e)}

    # ---- parity: closed form over ALL ranks' inputs
    expected_parts = []
    mine_expected = co.dlog_expected(g, scalars, seed_b, n)          # affine bytes of this rank's shard
    if world > 1:
        buf = [torch.empty(aff, dtype=torch.uint8, device=cdev) for _ in range(world)]
        dist.all_gather(buf, torch.frombuffer(bytearray(mine_expected), dtype=torch.uint8).to(cdev))
        expected_parts = [t.cpu().numpy().tobytes() for t in buf]
    else:
        expected_parts = [mine_expected]
    bit_exact = None
    if rank == 0:
        # lift each shard's expected affine point to Jacobian (x, y, 1) and add them up
        zc = _mont_one() if g == "g1" else _mont_one() + bytes(48)   # Z = 1 (Fp) or 1 + 0u (Fp2)
        jac = b"".join((e + zc) if e != bytes(aff) else bytes(aff + len(zc)) for e in expected_parts)
        want = co.to_affine(g, co.sum_jac(g, jac, world))
        bit_exact = co.to_affine(g, result) == want

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        total_points = n * world
        value = total_points / (elapsed / args.steps)
        acc_ms = sum(p["accumulate_ms"] for p in prof_acc) / len(prof_acc)
        p0 = prof_acc[-1]
        alg_bytes = ALG_BYTES_PER_POINT[g] * n
        achieved_gbs = alg_bytes / (acc_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel: measured with rocprofv3 PMC passes (tools/profile_bench.sh) on this exact
        # workload and committed under profiles/; bench.py cannot collect counters itself.
        traffic, traffic_src = None, None
        try:
            key = f"msmk::k_accumulate<msmk::{g.upper()}C>"
            if args.log_n == 20:   # the committed counters were collected on this exact workload
                for f in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_summary.json")), reverse=True):
                    pj = json.load(open(os.path.join(ROOT, "profiles", f)))
                    if key in pj.get("kernels", {}) and "hbm_bytes_per_launch_corrected" in pj["kernels"][key]:
                        traffic = pj["kernels"][key]["hbm_bytes_per_launch_corrected"]
                        traffic_src = "profiles/" + f
                        break
        except Exception:
            pass
        out = {
            "metric": f"{g.upper()} MSM points/sec",
            "value": value,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "bit_exact": bit_exact,
            "config": {"workload": f"{g.upper()} MSM, 2^{args.log_n} random bases+scalars per GPU, bases resident, scalars in HBM",
                       "points_per_gpu": n, "total_points": total_points, "window_bits": p0["window_bits"],
                       "num_windows": p0["num_windows"], "parallelism": f"base-set sharded x{world}", "scalar_dist": args.dist, "host_threads_issuing": args.concurrency,
                       "field_repr": "14 x 28-bit limbs in u32, products accumulated with v_mad_u64_u32"},
            "roofline": {"bound": "hbm", "kernel": f"k_accumulate<{g.upper()}C>", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": acc_ms,
                         "note": "the bucket method re-reads each 128-B device point once per window (16x at c=16): "
                                 "traffic ~ 16 x algorithmic by design; the kernel is VALU-bound, see valu_roofline"},
            "valu_roofline": {"model_mads_per_point": MADS_PER_POINT[g],
                              "achieved_Tmad_s": MADS_PER_POINT[g] * n / (acc_ms * 1e-3) / 1e12,
                              "peak_Tmad_s": MAD_PEAK_TLOPS,
                              "frac": MADS_PER_POINT[g] * n / (acc_ms * 1e-3) / 1e12 / MAD_PEAK_TLOPS,
                              "note": "integer VALU (v_mad_u64_u32) is the real bound of this path; HBM frac is low by construction"},
            "phases_ms": {k: sum(p[k] for p in prof_acc) / len(prof_acc) for k in
                          ("digits_ms", "scan_ms", "scatter_ms", "accumulate_ms", "reduce_ms", "combine_ms", "d2h_ms", "host_fold_ms", "total_ms")},
            "input_gen_s": gen_s,
            "two_host_threads": two_thread,
            "pairing_2p16": pairing,
            "call_shapes": call_shapes,
        }
        if not args.no_cpu_baseline and world == 1:   # reported on rank 0 at N = 1 only
            # one warm-up + median of >= 5 runs (3 at 2^24 and above), CPU model and core count in the result,
            # plus a single-thread figure (on a 2^16-point prefix) for scaling
            import statistics
            runs = 5 if args.log_n <= 20 else (3 if args.log_n <= 24 else 1)
            if args.log_n <= 22:
                co.msm(g, bases, scalars, n, 0, ncpu)
            times = []
            for _ in range(runs):
                t0 = time.perf_counter()
                cpu = co.msm(g, bases, scalars, n, 0, ncpu)
                times.append(time.perf_counter() - t0)
            med = statistics.median(times)
            assert co.to_affine(g, cpu) == co.dlog_expected(g, scalars, seed_b, n)
            n1 = min(n, 1 << 16)
            t0 = time.perf_counter()
            co.msm(g, bases[:aff * n1], scalars[:32 * n1], n1, 0, 1)
            t_single = time.perf_counter() - t0
            model = "unknown"
            try:
                for line in open("/proc/cpuinfo"):
                    if line.startswith("model name"):
                        model = line.split(":", 1)[1].strip()
                        break
            except OSError:
                pass
            out["cpu_baseline"] = {"value": n / med, "unit": "points/s", "cores": ncpu, "kind": "port", "cpu_model": model,
                                   "sample": f"full workload of one GPU (2^{args.log_n} points), median of {runs} runs after a warm-up, "
                                             "blst-style Pippenger restatement in portable C (oracle/msm_oracle.c), not blst assembly",
                                   "seconds": med, "best_seconds": min(times),
                                   "single_thread": {"value": n1 / t_single, "unit": "points/s", "points": n1}}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def _host_threads() -> int:
    """Threads for the CPU legs: the cgroup CPU quota when there is one (a 1-GPU box is given ~16 cores of a
    256-thread host), else the affinity mask."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except Exception:
        pass
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    return min(n, 16)


def _mont_one() -> bytes:
    """Montgomery form of 1 (R mod p), /root/reference/src/fp.rs:532 — Z coordinate of an affine point lifted to Jacobian."""
    limbs = [0x760900000002FFFD, 0xEBF4000BC40C0002, 0x5F48985753C758BA, 0x77CE585370525745, 0x5C071A97A256EC6D, 0x15F65EC3FA80E493]
    return b"".join(l.to_bytes(8, "little") for l in limbs)


if __name__ == "__main__":
    main()
