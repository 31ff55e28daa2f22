#!/usr/bin/env python3
"""bench.py — G1 MSM points/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--log-n L] [--group g1|g2]
  N > 1:  either the bare command above (bench.py then starts its N ranks itself: a child `python -m torch.distributed.run`
          spawned BEFORE the parent touches the GPU; the parent relays the child's output and exit code), or the launcher form
          python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
              bench.py --gpus N --steps K --warmup W

A "step" is one full MSM over one batch of synthetic (base, scalar) pairs through the C ABI (mi_msm_g1_device): digit
extraction, bucket sort, bucket accumulation, bucket reduction, per-window combine, host Horner fold — nothing is cached
between steps.  Inputs are resident in HBM when the timed region starts (bases as the resident SRS, scalars in a device
buffer).

  N = 1   BASELINE config #2: 2^20 points (override with --log-n).  The same JSON line carries, as `secondary`, the other
          single-GPU configs of BASELINE.json — G1 2^24 (the north-star size), G2 2^20 (config #4), 2^16 pairs of
          Miller loop + final exponentiation (config #5) — each with its own parity flag, roofline and (pairing) CPU baseline.
  N > 1   BASELINE config #3: 2^24 points IN TOTAL, the base set sharded contiguously, 2^24 / N per rank (strong scaling;
          --log-n L switches to 2^L per GPU, weak).  No data-path collective; each step ends with the RCCL all-gather of
          the N 144-byte partial sums and the deterministic fold on every rank.

The result of the last step is checked bit-exact (canonical affine bytes) against the closed form (sum s_i k_i) G.
The oracle (oracle/) is used ONLY to generate the synthetic inputs, as the checker, and as the timed CPU baseline
(`cpu_baseline`, kind "port": the reference's blst path cannot be built in this image).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED_B, SEED_S = 0xA55E7 + 2, 0x5CA1A5 + 2
ALG_BYTES_PER_POINT = {"g1": 128, "g2": 224}   # base + scalar, each read once
FP_MULS_PER_ADD = {"g1": 10, "g2": 30}         # XYZZ mixed addition = 10 field multiplications; an Fp2 one = 3 Fp ones
MADS_PER_FP_MUL = 300                          # 12^2 + 12^2 + 12 32-bit multiply-adds
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: 8 TB/s spec
MAD_PEAK_T = 39.3                              # 1024 SIMD x 64 lanes x 2.4 GHz / 4 cycles (v_mad_u64_u32 theoretical issue rate)
MAD_MEASURED_T = 33.4                          # tools/ubench_valu.hip on MI355X: what a pure MAD loop reaches (profiles/r01_ubench_valu.txt)
# k_accumulate<G1C>: instructions of one mixed addition in the shipped code object (tools/kernel_resources.py + llvm-objdump) priced
# with the measured per-instruction costs at two waves per SIMD (profiles/r03_ubench_carry.txt): 3542 v_mad_u64_u32 x 4.8 + 126 v_mul_lo x 4.4
# + 234 v_lshrrev_b64 x 4.6 + 235 v_lshl_add_u64 x 5.05 + 257 v_and x 2.6 + ~330 others x 2.6 (DESIGN_HISTORY.md §9)
SIMDS, CLOCK_HZ = 1024, 2.4e9
# Sum of the measured per-instruction issue costs (real cycles at the measured shader clock; tools/ubench_carry.hip re-based with
# tools/probe/clock_probe.hip: clock64 ticks 100 MHz, the figures of profiles/r03_ubench_*.txt are 2.4-GHz pseudo-cycles and shrink by
# clock / 2.4) of ONE wave-wide mixed addition, counted off the shipped code objects by tools/kernel_resources.py --mix; filled from
# the newest profiles/r*_instruction_costs.json when present
def _load_instruction_costs():
    """the newest profiles/r*_instruction_costs.json (tools/instruction_costs.py: hot loops of the shipped code objects priced per instruction)"""
    try:
        pdir = os.path.join(ROOT, "profiles")
        f = sorted(f for f in os.listdir(pdir) if f.endswith("_instruction_costs.json"))[-1]
        return json.load(open(os.path.join(pdir, f)))
    except Exception:
        return {}
ADD_INSTRUCTION_COST = _load_instruction_costs()
PAIRING_FP_MULS_PER_PAIR = 63 * (31 + 39) + 5 * (41 + 39)   # DESIGN_HISTORY.md §5: line + sparse Fp12 product per step, squarings shared


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _under_profiler() -> bool:
    """rocprofv3 preloads its tool library into the process (and with --pmc initialises the GPU before main() runs): such a process
    must not start child processes that use the GPU.  Profiling is single-process: tools/profile_bench.sh passes --no-secondary."""
    env = os.environ
    return any(k.startswith("ROCPROF") for k in env) or "rocprof" in env.get("LD_PRELOAD", "") or "rocprofiler" in env.get("HSA_TOOLS_LIB", "")


def _self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as ONE child process tree (`python -m
    torch.distributed.run`, one rank per GPU) and relay its output and exit code.  The parent has not imported torch and never
    touches the GPU (a process that has initialised the GPU must not be replaced or re-executed on this pool)."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


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


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _mont_one() -> bytes:
    """Montgomery form of 1 (R mod p), /root/reference/src/fp.rs:532 — Z coordinate of an affine point lifted to Jacobian."""
    limbs = [0x760900000002FFFD, 0xEBF4000BC40C0002, 0x5F48985753C758BA, 0x77CE585370525745, 0x5C071A97A256EC6D, 0x15F65EC3FA80E493]
    return b"".join(l.to_bytes(8, "little") for l in limbs)


def source_hash() -> str:
    """sha256 over the sources every kernel is built from (ark-blst_amd/csrc/* and include/*.h, names and contents, sorted).  A profile
    summary under profiles/ carries the hash of the tree it was measured on (tools/summarize_profile.py); bench.py compares it with the
    tree it runs from and marks a replayed traffic figure `traffic_stale` when the two differ.  A hash of the
    sources, not of the .so: the library is rebuilt on other boxes, and byte-identical output is not promised."""
    import hashlib
    h = hashlib.sha256()
    for d in (os.path.join(ROOT, "ark-blst_amd", "csrc"), os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".cuh", ".hpp", ".h")) or f == "Makefile":
                h.update(f.encode())
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def _traffic(g: str, log_n, precomputed: bool):
    """HBM bytes per launch of the accumulate kernel: measured with rocprofv3 PMC passes (tools/profile_bench.sh) on this
    exact workload and committed under profiles/ (bench.py cannot collect counters itself); (None, None, None) when no summary matches.
    Third value: True when the summary was measured on other sources than the ones this run was built from."""
    try:
        key = "msmk::k_accumulate<msmk::G1C>" if g == "g1" else "msmk::k_accumulate_g2_coop<msmk::G2C>"
        files = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_summary.json")), reverse=True)
        for f in files:
            pj = json.load(open(os.path.join(ROOT, "profiles", f)))
            wl = pj.get("workload")
            if wl is None:   # round-1 summaries carry no workload key: the r01 MSM summaries are the default line
                wl = {"group": "g1", "log_n": 20, "precomputed": False} if "_msm_" in f or f.startswith("r01_d_") else {}
            if (wl.get("group"), wl.get("log_n"), bool(wl.get("precomputed"))) != (g, log_n, precomputed):
                continue
            if key in pj.get("kernels", {}) and "hbm_bytes_per_launch_corrected" in pj["kernels"][key]:
                return pj["kernels"][key]["hbm_bytes_per_launch_corrected"], "profiles/" + f, pj.get("source_sha256") != source_hash()
    except Exception:
        pass
    return None, None, None


def _rows_f_traffic(g: str, log_n: int, kernel_keys):
    """HBM bytes of the named kernels of a rows-(f) call (their LARGEST launch = the full-size call; warm-up launches are smaller) from the newest
    committed PMC summary of tools/profile_rows_f.sh for this group and size: (bytes, source, stale) or (None, None, None)."""
    try:
        files = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if "rows_f" in f and f.endswith("_pmc_summary.json")), reverse=True)
        for f in files:
            pj = json.load(open(os.path.join(ROOT, "profiles", f)))
            wl = pj.get("workload") or {}
            if (wl.get("group"), wl.get("log_n")) != (g, log_n):
                continue
            ks = pj.get("kernels", {})
            tot, found = 0.0, 0
            for key in kernel_keys:
                # a kernel that runs in several launches per call (chunks; the tree's levels): launches per call x mean bytes per launch
                for k, v in ks.items():
                    if key in k and "hbm_bytes_per_launch_corrected" in v:
                        tot += v["hbm_bytes_per_launch_corrected"] * v.get("launches", 1) / max(1, pj.get("calls_profiled", 1))
                        found += 1
            if found:
                return tot, "profiles/" + f, pj.get("source_sha256") != source_hash()
    except Exception:
        pass
    return None, None, None


def _pairing_traffic():
    """HBM bytes of the two Miller kernels (largest launch of each = the 2^16-pair call) from the newest committed PMC summary."""
    try:
        files = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if "pairing" in f and f.endswith("_pmc_summary.json")), reverse=True)
        for f in files:
            pj = json.load(open(os.path.join(ROOT, "profiles", f)))
            ks = pj.get("kernels", {})
            t = [v["hbm_bytes_largest_launch_corrected"] for k, v in ks.items()
                 if ("k_miller_lines2" in k or "k_miller_accumulate" in k) and "hbm_bytes_largest_launch_corrected" in v]
            if len(t) == 2:
                return sum(t), "profiles/" + f, pj.get("source_sha256") != source_hash()
    except Exception:
        pass
    return None, None, None


def _expected_ms(g: str, log_n):
    """single-GPU ms per call at this shard size from the newest committed size sweep (profiles/*_sweep_<g>_*.jsonl): what one
    rank of a sharded run should take before any exchange cost.  The plan's own pick (forced_c = 0), its warm row (`auto_row: last`,
    tools/sweep_sizes.py) where the file has one."""
    try:
        files = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f"_sweep_{g}_" in f and f.endswith(".jsonl")), reverse=True)
        for f in files:
            rows = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", f))]
            rows = [r for r in rows if r.get("log_n") == log_n and r.get("ok", True) and not r.get("forced_c")]
            if rows:
                rows.sort(key=lambda r: r.get("auto_row") != "last")
                return rows[0]["ms"], "profiles/" + f
    except Exception:
        pass
    return None, None


def _distort(scalars: bytes, n: int, dist: str) -> bytes:
    if dist == "uniform":
        return scalars
    import numpy as np
    a = np.frombuffer(scalars, dtype=np.uint8).reshape(n, 32).copy()
    if dist == "zero_one":       # R1CS-like witness: bits
        a[:, 1:] = 0
        a[:, 0] &= 1
    elif dist == "small64":      # 64-bit values
        a[:, 8:] = 0
    elif dist == "r1cs_mix":     # witness-like: half zeros, a quarter ones, a quarter full-size values
        sel = a[:, 0] & 3
        a[sel < 2] = 0
        a[sel == 2] = 0
        a[sel == 2, 0] = 1
    elif dist == "all_ones":     # every scalar = 1: the plain sum of the bases, one bucket of N entries
        a[:] = 0
        a[:, 0] = 1
    else:                        # every scalar identical: all points in one bucket per window
        a[:] = a[0]
    return a.tobytes()


def _measured_clock(kernel_key: str, g: str = None, log_n=None):
    """Shader clock of the timed kernel: GRBM_GUI_ACTIVE cycles per launch (a committed rocprofv3 PMC pass) / the kernel's average
    duration in the kernel trace of the SAME profile run (tools/summarize_profile.py writes both; the r03 summaries carry the cycles
    and their kernel_stats.csv the duration).  Returns (GHz, source) or (None, None)."""
    try:
        import csv
        pdir = os.path.join(ROOT, "profiles")
        for f in sorted((f for f in os.listdir(pdir) if f.endswith("_pmc_summary.json")), reverse=True):
            pj = json.load(open(os.path.join(pdir, f)))
            wl = pj.get("workload") or {}
            if g is not None and (wl.get("group"), wl.get("log_n")) != (g, log_n):
                continue
            for k, v in pj.get("kernels", {}).items():
                if kernel_key not in k:
                    continue
                if "shader_clock_ghz" in v:
                    return v["shader_clock_ghz"], "profiles/" + f
                if "gpu_cycles_per_launch" not in v:
                    continue
                stats = os.path.join(pdir, f.replace("_pmc_summary.json", "_kernel_stats.csv"))
                for r in csv.DictReader(open(stats)):
                    if kernel_key in r["Name"]:
                        return v["gpu_cycles_per_launch"] / float(r["AverageNs"]), "profiles/" + f + " + " + os.path.basename(stats)
    except Exception:
        pass
    return None, None


def _valu_roofline(kernel: str, model: str, mads_per_launch: float, kernel_ms: float, clock, clock_src, extra: dict) -> dict:
    """The binding roofline of this path: integer multiply-add issue.  peak = 1024 SIMDs x 64 lanes / 4 cycles per
    wave64 v_mad_u64_u32 x 2.4 GHz (the guide's peak clock) = 39.3 T MAD/s; the kernels run at a lower clock under this load, so the
    same bound at the MEASURED shader clock is printed beside it."""
    tmad = mads_per_launch / (kernel_ms * 1e-3) / 1e12
    r = {"bound": "valu_int_mad", "kernel": kernel, "achieved": tmad, "peak": MAD_PEAK_T, "unit": "T MAD/s", "frac": tmad / MAD_PEAK_T,
         "model": model, "model_mads_per_launch": mads_per_launch, "kernel_ms": kernel_ms,
         "measured_peak": MAD_MEASURED_T, "frac_of_measured_peak": tmad / MAD_MEASURED_T,
         "shader_clock_ghz": clock, "shader_clock_source": clock_src}
    if clock:
        peak_clk = SIMDS * 64 * clock * 1e9 / 4 / 1e12
        r["peak_at_measured_clock"] = peak_clk
        r["frac_at_measured_clock"] = tmad / peak_clk
    r.update(extra)
    return r


MAX_LINE_BYTES = 8000   # the headline JSON line (a 40 KB line left BENCH_r04.json unparsed)
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_stale", "algorithmic_bytes_per_launch",
              "kernel_ms", "model", "model_mads_per_point", "shader_clock_ghz", "frac_at_measured_clock", "instruction_cost_sum_cycles",
              "cycles_per_wave_addition", "frac_of_instruction_cost_bound")
_HBM_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel_ms")


def _compact(d: dict, keys) -> dict:
    return {k: d[k] for k in keys if k in d}


def _clock_of(profs):
    """Shader clock of the accumulate kernels of the timed steps, measured INSIDE the kernel (mi_profile.accumulate_clock_ghz: the
    ratio of the waves' s_memtime and s_memrealtime sums): (GHz, source) or (None, None) when no step carried it."""
    t = sum(p.get("accumulate_clock_ticks", 0) for p in profs)
    r = sum(p.get("accumulate_ref_ticks", 0) for p in profs)
    if r:
        return 0.1 * t / r, "measured in this run: s_memtime / s_memrealtime sums of the timed accumulate launches (mi_profile.accumulate_clock_ghz)"
    return None, None


def _rooflines(g: str, n: int, log_n, acc_ms: float, nwin: int, precomputed: bool, clock_now=None) -> dict:
    alg_bytes = ALG_BYTES_PER_POINT[g] * n
    gbs = alg_bytes / (acc_ms * 1e-3) / 1e9
    traffic, src, stale = _traffic(g, log_n, precomputed) if log_n is not None else (None, None, None)
    mads = nwin * FP_MULS_PER_ADD[g] * MADS_PER_FP_MUL       # window-aware: one mixed addition per point and window
    kernel = "k_accumulate<G1C>" if g == "g1" else "k_accumulate_g2_coop<G2C>"
    key = "k_accumulate<msmk::G1C>" if g == "g1" else "k_accumulate_g2_coop"
    clock, clock_src = clock_now if clock_now and clock_now[0] else _measured_clock(key, g, log_n)   # this run's own clock; a committed profile's only as a fallback
    if clock is None:
        clock, clock_src = _measured_clock(key)
    # real cycles one SIMD spends per wave-wide mixed addition (64 additions), incl. the per-bucket set-up and the kernel's tail
    wave_adds = nwin * n / 64.0
    cyc = lambda hz: acc_ms * 1e-3 * hz * SIMDS / wave_adds
    cost = ADD_INSTRUCTION_COST.get(g)
    extra = {"traffic": traffic, "traffic_source": src, "traffic_stale": stale, "algorithmic_bytes_per_launch": alg_bytes,
             "model_mads_per_point": mads,
             "cycles_per_wave_addition": cyc(clock * 1e9) if clock else None,
             "cycles_per_wave_addition_at_2p4ghz": cyc(CLOCK_HZ),
             "instruction_cost_sum_cycles": cost["cycles"] if cost else None,
             "instruction_cost_source": cost["source"] if cost else None,
             "frac_of_instruction_cost_bound": (cost["cycles"] / cyc(clock * 1e9)) if (cost and clock) else None,
             "note": "integer VALU (v_mad_u64_u32) is the bound of this path.  frac prices the MODEL's multiply-adds (300 per field "
                     "multiplication on 32-bit limbs) at 4 cycles and 2.4 GHz; the kernel executes 28-bit limbs (406 multiply-adds per "
                     "multiplication, no carry instructions) at the measured shader clock: instruction_cost_sum_cycles is the sum of its "
                     "instructions' measured issue costs per wave-wide addition, frac_of_instruction_cost_bound how close it runs to that"}
    valu = _valu_roofline(kernel, f"{nwin} windows x {FP_MULS_PER_ADD[g]} Fp-mul x {MADS_PER_FP_MUL} MAD per point", mads * n, acc_ms, clock, clock_src, extra)
    return {
        "roofline": valu,
        "hbm_roofline": {"bound": "hbm", "kernel": kernel, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src, "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_ms": acc_ms,
                         "note": "reported because the north star asks for it; low by construction (the bucket method re-reads each "
                                 "device point once per window: traffic ~ windows x algorithmic bytes, still far below 8 TB/s)"},
    }


class MsmLeg:
    """One MSM workload on this rank's GPU: inputs, resident bases, device scalars, timed steps, parity."""

    def __init__(self, pkg, co, torch, g, n, seed_b, seed_s, ncpu, device, dist="uniform", window_bits=0, precomputed=False, validate=True):
        self.pkg, self.co, self.torch, self.g, self.n = pkg, co, torch, g, n
        self.seed_b, self.seed_s, self.ncpu = seed_b, seed_s, ncpu
        t0 = time.time()
        self.bases = co.gen_bases(g, seed_b, n, ncpu)
        self.scalars = _distort(co.gen_scalars(seed_s, n), n, dist)
        self.gen_s = time.time() - t0
        self.ctx = pkg.Context([device])
        if window_bits and not precomputed:
            self.ctx.set_window_bits(window_bits)
        t0 = time.time()
        if precomputed:
            self.ctx.set_bases_precomputed(g, self.bases, n, window_bits)
        else:
            self.ctx.set_bases(g, self.bases, n)
        self.set_bases_s = time.time() - t0
        # An SRS is validated ONCE when it is loaded (Valid::check of every point, as arkworks' deserialisers do with Validate::Yes): the
        # library does it on the GPU and records a clean set, which lets the MSMs over it fold the scalars' signs (one digit window
        # fewer at c = 15 / 17 — nothing at the 2^20 and 2^24 sizes, ~5 % at the 2^21-2^23 shards of config #3).  Outside every timed region.
        t0 = time.time()
        self.validated = validate and self.ctx.validate_bases(g) == 0
        self.validate_s = time.time() - t0
        self.d_scalars = torch.frombuffer(bytearray(self.scalars), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()   # the library reads the scalars on its own stream (include/arkblst_amd.h)

    def call(self) -> bytes:
        return self.ctx.msm_device(self.g, self.d_scalars.data_ptr(), self.n, self.pkg.SCALAR_CANONICAL)

    def expected_affine(self) -> bytes:
        return self.co.dlog_expected(self.g, self.scalars, self.seed_b, self.n)

    def run(self, steps: int, warmup: int) -> dict:
        for _ in range(warmup):
            self.call()
        self.torch.cuda.synchronize()
        profs, result = [], b""
        t0 = time.perf_counter()
        for _ in range(steps):
            result = self.call()
            profs.append(self.ctx.profile_raw())   # the C struct only: the dicts are built after the timed region
        self.torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        return {"elapsed": elapsed, "result": result, "profs": [self.pkg.profile_dict(p) for p in profs]}

    def phase_breakdown(self, calls: int = 3) -> list:
        """A few more calls OUTSIDE any timed region with every phase's events recorded (mi_msm_set_profile_level 2): the timed steps
        record the accumulate kernel's interval only — each event record idles the device for ~6 us between two kernels."""
        self.ctx.set_profile_level(2)
        try:
            out = []
            for _ in range(calls):
                self.call()
                out.append(self.ctx.profile())
            return out
        finally:
            self.ctx.set_profile_level(1)

    def close(self):
        self.ctx.close()
        self.d_scalars = None


def _phases(profs, breakdown=None) -> dict:
    """Phase times per call: accumulate / host fold / total from the TIMED steps' profiles, the other phases from the breakdown calls
    made after the timed region (MsmLeg.phase_breakdown)."""
    keys = ("digits_ms", "scan_ms", "scatter_ms", "accumulate_ms", "reduce_ms", "combine_ms", "d2h_ms", "host_fold_ms", "total_ms")
    timed = ("accumulate_ms", "host_fold_ms", "total_ms")
    src = breakdown or profs
    return {k: sum(p[k] for p in (profs if k in timed else src)) / len(profs if k in timed else src) for k in keys}


ALG_AFF = {"g1": 96, "g2": 192}


def _secondary_msm(pkg, co, torch, g, log_n, seed_off, ncpu, device, steps, precomputed=False, window_bits=0, cpu_sample_log_n=0) -> dict:
    n = 1 << log_n
    leg = MsmLeg(pkg, co, torch, g, n, SEED_B + seed_off, SEED_S + seed_off, ncpu, device, precomputed=precomputed, window_bits=window_bits)
    try:
        r = leg.run(steps, 1)
        ok = co.to_affine(g, r["result"]) == leg.expected_affine()
        ph = _phases(r["profs"], leg.phase_breakdown(3 if log_n <= 22 else 1))
        p0 = r["profs"][-1]
        out = {"metric": f"{g.upper()} MSM points/sec", "value": n * steps / r["elapsed"], "unit": "points/s", "ms_per_step": r["elapsed"] / steps * 1e3,
               "steps": steps, "bit_exact": ok, "workload": f"{g.upper()} MSM, 2^{log_n} random bases+scalars, bases resident"
               + (" as precomputed 2^(c j) P tables" if precomputed else "") + ", scalars in HBM",
               "window_bits": p0["window_bits"], "num_windows": p0["num_windows"], "phases_ms": ph, "input_gen_s": leg.gen_s,
               "set_bases_s": leg.set_bases_s}
        out.update(_rooflines(g, n, log_n, ph["accumulate_ms"], p0["num_windows"], precomputed, _clock_of(r["profs"])))
        out["kernel_ms"] = ph["accumulate_ms"]
        out["step_frac"] = out["roofline"]["model_mads_per_point"] * n / (out["ms_per_step"] * 1e-3) / 1e12 / MAD_PEAK_T
        out["window_groups"] = p0.get("window_groups", 1)
        if cpu_sample_log_n:   # the CPU port beside the north-star size, on a bounded prefix (the full 2^24 workload takes the port a minute)
            m = 1 << cpu_sample_log_n
            t1 = time.perf_counter()
            cpu = co.msm(g, leg.bases[:ALG_AFF[g] * m], leg.scalars[:32 * m], m, 0, ncpu)
            cpu_s = time.perf_counter() - t1
            assert co.to_affine(g, cpu) == co.dlog_expected(g, leg.scalars[:32 * m], leg.seed_b, m)
            out["cpu_baseline"] = {"value": m / cpu_s, "unit": "points/s", "cores": ncpu, "kind": "port", "cpu_model": _cpu_model(), "seconds": cpu_s,
                                   "sample": f"the first 2^{cpu_sample_log_n} of the 2^{log_n} points, one run (blst-style Pippenger restatement in C, oracle/msm_oracle.c)"}
        return out
    finally:
        leg.close()


def _pairing_leg(pkg, co, ncpu, device) -> dict:
    """BASELINE config #5: 2^16 G1 x G2 pairs through mi_multi_pairing.  Parity: prod e(P_i, Q_i) e(-P_i, Q_i) == 1 at full size
    and 256 random pairs bit-exact against the C oracle, which is also the timed CPU baseline."""
    from oracle import pairing as pr_oracle, bls12_381 as o
    half = 1 << 15
    p1 = co.gen_bases("g1", SEED_B + 101, half, ncpu)
    q2 = co.gen_bases("g2", SEED_B + 102, half, ncpu)
    neg = bytearray(p1)
    for i in range(half):   # -P: y -> p - y (Montgomery form, y != 0)
        y = int.from_bytes(p1[96 * i + 48:96 * i + 96], "little")
        neg[96 * i + 48:96 * i + 96] = (o.P - y).to_bytes(48, "little")
    P_all, Q_all = p1 + bytes(neg), q2 + q2
    n = 2 * half
    with pkg.Context([device]) as ctx:
        ctx.multi_pairing(P_all[:96 * 64], Q_all[:192 * 64])
        best, pp = 1e30, None
        for _ in range(3):
            t1 = time.perf_counter()
            gt = ctx.multi_pairing(P_all, Q_all)
            dt = time.perf_counter() - t1
            if dt < best:
                best, pp = dt, ctx.pairing_profile()
        m = 4096   # CPU baseline sample: ~1 s on 16 host threads (the final exponentiation is then < 1 % of it)
        t1 = time.perf_counter()
        cpu_gt = co.multi_pairing(p1[:96 * m], q2[:192 * m], ncpu)
        cpu_s = time.perf_counter() - t1
        sample_ok = ctx.multi_pairing(p1[:96 * m], q2[:192 * m]) == cpu_gt
    acc_ms, lines_ms, miller_ms = pp["accumulate_ms"], pp["lines_ms"], pp["miller_ms"]
    mads = PAIRING_FP_MULS_PER_PAIR * MADS_PER_FP_MUL
    gbs = 288.0 * n / (miller_ms * 1e-3) / 1e9
    traffic, tsrc, tstale = _pairing_traffic()
    clock, clock_src = _measured_clock("k_miller_accumulate")
    pc = ADD_INSTRUCTION_COST.get("pairing") or {}
    valu = _valu_roofline("k_miller_lines2 + k_miller_accumulate",
                          f"{PAIRING_FP_MULS_PER_PAIR} Fp-mul per pair (63 doubling + 5 addition steps: line + sparse Fp12 product; "
                          f"Fp12 squarings shared by all pairs) x {MADS_PER_FP_MUL} MAD", mads * n, miller_ms, clock, clock_src,
                          {"traffic": traffic, "traffic_source": tsrc, "traffic_stale": tstale, "algorithmic_bytes_per_launch": 288 * n, "model_mads_per_pair": mads,
                           "per_kernel": {"k_miller_lines2": {"ms": lines_ms, "model_Tmad_s": (63 * 31 + 5 * 41) * MADS_PER_FP_MUL * n / (lines_ms * 1e-3) / 1e12},
                                          "k_miller_accumulate": {"ms": acc_ms, "model_Tmad_s": 68 * 39 * MADS_PER_FP_MUL * n / (acc_ms * 1e-3) / 1e12}},
                           "instruction_cost": pc or None})
    return {"metric": "pairs/s, batched Miller loop + final exponentiation (host buffers in, Gt out)", "value": n / best, "unit": "pairs/s",
            "n_pairs": n, "ms": best * 1e3, "miller_kernels_ms": miller_ms, "k_miller_lines2_ms": lines_ms,
            "k_miller_accumulate_ms": acc_ms, "pairs_per_accumulator": pp["pairs_per_accumulator"], "fp12_tree_ms": pp["tree_ms"], "h2d_ms": pp["h2d_ms"],
            "host_tail_ms": pp["host_ms"],
            "product_cancels_to_one": gt == pr_oracle.fp12_to_bytes(pr_oracle.FP12_ONE), f"bit_exact_{m}_pairs_vs_c_oracle": sample_ok,
            "bit_exact": sample_ok,
            "roofline": valu,
            "hbm_roofline": {"bound": "hbm", "kernel": "k_miller_lines2 + k_miller_accumulate", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc, "algorithmic_bytes_per_launch": 288 * n, "kernel_ms": miller_ms,
                             "note": "288 B per pair in (96 B G1 + 192 B G2 affine); the 26 KB of line coefficients per pair written and "
                                     "re-read between the two kernels are counted as traffic, not as algorithmic bytes"},
            "cpu_baseline": {"value": m / cpu_s, "unit": "pairs/s", "cores": ncpu, "kind": "port", "cpu_model": _cpu_model(),
                             "sample": f"{m} pairs incl. one final exponentiation; textbook affine Miller loop in portable C "
                                       "(oracle/pairing_oracle.c), an order of magnitude slower per core than assembly libraries", "seconds": cpu_s}}


NORM_FP_MULS_PER_POINT = 12          # product tree up + down (fan-out 8): ~3.5, Z^-2, Z^-3, x and y: 4, conversions in and out: 5
# G1 decoder as the kernels of round 6 execute it (a squaring counts as one multiplication): y = (x^3 + 4)^((p+1)/4) by a sliding window of four
# bits, 378 squarings + 87 products, + 5 around it = 470; subgroup test: two 63-step Jacobian ladders, 7 per doubling, 5 mixed additions of 11 and 5
# general ones of 16, + 4 for the verdict = 1021.  (Rounds 2-5: square-and-multiply 570 + complete projective formulas 1650.)
DESER_FP_MULS_PER_POINT = 470 + 1021


# G2 decoder (round 6): Fp2 square root by the complex method, two Fp exponentiations by (p-3)/4 on the sliding window (378 squarings + 87
# products each) = 930, + ~20 around them; subgroup test: one 63-step Jacobian ladder over Fp2, 4 squarings (2 Fp-mul) + 3 products (3 Fp-mul) = 17
# per doubling, 5 mixed additions of 29, + ~15 for psi and the verdict = 1231.  (Rounds 2-5: two Fp2 exponentiations 2656 + complete formulas 1566.)
DESER_G2_FP_MULS_PER_POINT = 950 + 1231


def _normalize_leg(pkg, co, ncpu, device, g="g1", log_n=20) -> dict:
    """Row (f)-2 of SURVEY.md 8: CurveGroup::normalize_batch (src/g1.rs:537-543, src/g2.rs:517-523) for 2^20 points with non-trivial Z
    through mi_g{1,2}_normalize_batch (host buffers in and out; the kernels are timed on the library's stream), all of them checked against
    the C oracle, which is also the timed CPU baseline."""
    n = 1 << log_n
    m = 1 << 12
    aff, jb, G = (96, 144, "G1") if g == "g1" else (192, 288, "G2")
    bases = co.gen_bases(g, SEED_B + 201, m + 1, ncpu)
    one = _mont_one() if g == "g1" else _mont_one() + bytes(48)
    jac = b"".join(co.sum_jac(g, bases[aff * i:aff * (i + 1)] + one + bases[aff * (i + 1):aff * (i + 2)] + one, 2) for i in range(m))
    blob = jac * (n // m)
    with pkg.Context([device]) as ctx:
        ctx.normalize_batch(g, blob)          # sizes the context's staging buffers (steady state: nothing is allocated per call)
        import numpy as np
        best, kms, py_ms = 1e30, None, None
        dst = np.zeros(n * aff, dtype=np.uint8)   # the caller's output vector, reused across calls (a fresh one adds its first-touch page faults)
        for _ in range(3):
            t1 = time.perf_counter()
            ctx.normalize_batch(g, blob, into=dst)
            dt = (time.perf_counter() - t1) * 1e3
            pr = ctx.profile()
            if pr["total_ms"] < best:           # total_ms: wall time of the C call (host pointers in, host pointers out)
                best, kms, py_ms = pr["total_ms"], pr["accumulate_ms"], dt
        out = dst.tobytes()
        # the same entirely in device memory (mi_g1_normalize_batch_device): what a caller that keeps its points on the GPU pays
        import torch
        d_in = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
        d_out = torch.empty(n * aff, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        dev_ms, kms_chunked = 1e30, kms
        for _ in range(3):
            ctx.normalize_batch_device(g, d_in.data_ptr(), n, d_out.data_ptr())
            pr = ctx.profile()
            if pr["total_ms"] < dev_ms:   # kernels_ms: the kernels at full size (the host-slice call launches level 0 once per PCIe chunk, under the copies)
                dev_ms, kms = pr["total_ms"], pr["accumulate_ms"]
        dev_ok = d_out.cpu().numpy().tobytes() == out
        del d_in, d_out
    t1 = time.perf_counter()
    cpu = co.normalize_batch(g, blob, ncpu)
    cpu_s = time.perf_counter() - t1
    ok = out == cpu and dev_ok
    muls = NORM_FP_MULS_PER_POINT * (1 if g == "g1" else 3)
    mads = muls * MADS_PER_FP_MUL
    clock, clock_src = _measured_clock("k_accumulate<msmk::G1C>")
    gbs = (jb + aff) * n / (kms * 1e-3) / 1e9
    traffic, tsrc, tstale = _rows_f_traffic(g, log_n, ("k_norm_",))
    return {"metric": f"{G} points/s, normalize_batch (Jacobian -> affine, one inversion), host slices in and out", "value": n / (best * 1e-3), "unit": "points/s", "n": n,
            "call_ms_host_buffers": best, "kernels_ms": kms, "kernels_ms_chunked_host_call": kms_chunked, "call_ms_device_buffers": dev_ms, "python_binding_wall_ms": py_ms, "bit_exact": ok,
            "workload": f"2^{log_n} {G} Jacobian points with non-trivial Z; value = the C call with host buffers in and out (PCIe-inclusive, what the trait's "
                        "caller pays); kernels_ms and the device-buffer call beside it",
            "roofline": {"bound": "hbm", "kernel": "k_norm_load + k_norm_up/down x levels + k_norm_final", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc, "traffic_stale": tstale,
                         "algorithmic_bytes_per_launch": (jb + aff) * n, "kernel_ms": kms,
                         "note": f"algorithmic bytes = {jb} B Jacobian in + {aff} B affine out per point; the product tree adds ~3 slots of "
                                 "intermediate values per point and level-0 element; arithmetic is 12 field multiplications per point, "
                                 "so neither roof is close: the row is bound by its five dependent passes over the data"},
            "valu_roofline": _valu_roofline("normalize kernels", f"{muls} Fp-mul x {MADS_PER_FP_MUL} MAD per point", mads * n, kms, clock, clock_src, {}),
            "cpu_baseline": {"value": n / cpu_s, "unit": "points/s", "cores": ncpu, "kind": "port", "cpu_model": _cpu_model(), "seconds": cpu_s,
                             "sample": f"the same 2^{log_n} points: Montgomery's simultaneous inversion per thread slice (oracle/msm_oracle.c "
                                       "normalize_batch), what blstrs batch_normalize does"}}


def _deserialize_g2_leg(pkg, co, ncpu, device, log_n=18) -> dict:
    """Row (f)-4 for G2 (src/g2.rs:366-411): 2^18 compressed 96-byte encodings with Valid::check on through mi_g2_deserialize_batch (two
    kernels: decoder, then the subgroup test).  Parity: every decoded point equals the point it was serialised from, and a sample is
    decoded by the C oracle, which is also the timed CPU baseline."""
    n = 1 << log_n
    bases = co.gen_bases("g2", SEED_B + 203, n, ncpu)
    with pkg.Context([device]) as ctx:
        enc = ctx.serialize_batch("g2", bases, True)
        import numpy as np
        into = (np.zeros(n * 192, dtype=np.uint8), np.zeros(n, dtype=np.uint8))   # the caller's vectors, reused (no first-touch page faults in the timed call)
        ctx.deserialize_batch("g2", enc, True, True, into=into)
        kms, wall = 1e30, 1e30
        for _ in range(2):
            ctx.deserialize_batch("g2", enc, True, True, into=into)
            if ctx.profile()["total_ms"] * 1e-3 < wall:
                kms, wall = ctx.profile()["accumulate_ms"], ctx.profile()["total_ms"] * 1e-3
        dec, st = into[0].tobytes(), into[1].tobytes()
        ctx.deserialize_batch("g2", enc, True, False, into=into)
        kms_novalidate = ctx.profile()["accumulate_ms"]
        t1 = time.perf_counter()
        rejected = ctx.set_bases_from_compressed("g2", enc, n, True, True)
        load_ms = (time.perf_counter() - t1) * 1e3
    m = 1 << 13   # CPU sample: ~1 s on 16 threads
    t1 = time.perf_counter()
    cdec, cst = co.g2_deserialize_batch(enc[:96 * m], True, True, 1, ncpu)
    cpu_s = time.perf_counter() - t1
    ok = dec == bases and st == bytes(n) and cdec == bases[:192 * m] and cst == bytes(m) and rejected == 0
    fp_muls = DESER_G2_FP_MULS_PER_POINT
    clock, clock_src = _measured_clock("k_accumulate<msmk::G1C>")
    return {"metric": "G2 points/s, deserialize_batch (compressed, validate on), host slices in and out", "value": n / wall, "unit": "points/s", "n": n,
            "call_ms_host_buffers": wall * 1e3, "kernel_ms": kms, "kernel_ms_validate_off": kms_novalidate, "set_bases_from_compressed_ms": load_ms, "bit_exact": ok,
            "workload": f"2^{log_n} compressed G2 encodings (96 B), decompression (Fp2 square root, complex method: two Fp exponentiations) + on-curve + subgroup check, host buffers in and out",
            "roofline": _valu_roofline("k_deserialize_g2 + k_validate_g2_coop", f"~{fp_muls} Fp-mul x {MADS_PER_FP_MUL} MAD per point",
                                       fp_muls * MADS_PER_FP_MUL * n, kms, clock, clock_src,
                                       dict(zip(("traffic", "traffic_source", "traffic_stale"), _rows_f_traffic("g2", log_n, ("k_deserialize_g2", "k_validate_g2_coop<1>"))),
                                            algorithmic_bytes_per_launch=(96 + 192) * n)),
            "cpu_baseline": {"value": m / cpu_s, "unit": "points/s", "cores": ncpu, "kind": "port", "cpu_model": _cpu_model(), "seconds": cpu_s,
                             "sample": f"{m} of the encodings: Fp2 square root by two Fp2 exponentiations + psi-endomorphism subgroup test in C "
                                       "(oracle/msm_oracle.c orc_g2_deserialize_batch, mode 1)"}}


def _deserialize_leg(pkg, co, ncpu, device, log_n=20) -> dict:
    """Row (f)-4: bulk G1 point decoding (src/g1.rs:386-431), 2^20 compressed encodings with Valid::check on, through
    mi_g1_deserialize_batch; all points checked against the inputs they were serialised from, a sample against the C oracle."""
    n = 1 << log_n
    bases = co.gen_bases("g1", SEED_B + 202, n, ncpu)
    with pkg.Context([device]) as ctx:
        enc = ctx.g1_serialize_batch(bases, True)
        import numpy as np
        into = (np.zeros(n * 96, dtype=np.uint8), np.zeros(n, dtype=np.uint8))   # the caller's vectors, reused (no first-touch page faults in the timed call)
        ctx.deserialize_batch("g1", enc, True, True, into=into)
        kms, wall = 1e30, 1e30
        for _ in range(2):
            ctx.deserialize_batch("g1", enc, True, True, into=into)
            if ctx.profile()["total_ms"] * 1e-3 < wall:   # total_ms: wall time of the C call
                kms, wall = ctx.profile()["accumulate_ms"], ctx.profile()["total_ms"] * 1e-3
        dec, st = into[0].tobytes(), into[1].tobytes()
        ctx.deserialize_batch("g1", enc, True, False, into=into)
        kms_novalidate = ctx.profile()["accumulate_ms"]
        # SRS loading as ONE call (decode + Valid::check + conversion on the GPU, only the 48-byte encodings cross PCIe) against the three
        # calls it replaces (deserialize to host, set_bases, validate_bases)
        ctx.set_bases_from_compressed("g1", enc, n, True, True)
        t1 = time.perf_counter()
        rejected = ctx.set_bases_from_compressed("g1", enc, n, True, True)
        load_ms = (time.perf_counter() - t1) * 1e3
        t1 = time.perf_counter()
        ctx.set_bases("g1", dec, n)
        bad = ctx.validate_bases("g1")
        three_ms = wall * 1e3 + (time.perf_counter() - t1) * 1e3
    m = 1 << 16   # CPU sample: ~1 s on 16 threads
    t1 = time.perf_counter()
    cdec, cst = co.g1_deserialize_batch(enc[:48 * m], True, True, 1, ncpu)
    cpu_s = time.perf_counter() - t1
    ok = dec == bases and st == bytes(n) and cdec == bases[:96 * m] and cst == bytes(m) and rejected == 0 and bad == 0
    mads = DESER_FP_MULS_PER_POINT * MADS_PER_FP_MUL
    clock, clock_src = _measured_clock("k_accumulate<msmk::G1C>")
    gbs = (48 + 96) * n / (kms * 1e-3) / 1e9
    return {"metric": "G1 points/s, deserialize_batch (compressed, validate on), host slices in and out", "value": n / wall, "unit": "points/s", "n": n,
            "call_ms_host_buffers": wall * 1e3, "kernel_ms": kms, "kernel_ms_validate_off": kms_novalidate,
            "srs_load": {"set_bases_from_compressed_ms": load_ms, "deserialize_then_set_bases_then_validate_ms": three_ms},
            "bit_exact": ok,
            "workload": f"2^{log_n} compressed G1 encodings (48 B), decompression + on-curve + subgroup check, host buffers in and out",
            "roofline": _valu_roofline("k_deserialize_g1", f"{DESER_FP_MULS_PER_POINT} Fp-mul (square root 470 + subgroup test 1021) x {MADS_PER_FP_MUL} MAD per point",
                                       mads * n, kms, clock, clock_src,
                                       dict(zip(("traffic", "traffic_source", "traffic_stale"), _rows_f_traffic("g1", log_n, ("k_deserialize_g1",))),
                                            algorithmic_bytes_per_launch=(48 + 96) * n)),
            "hbm_roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": (48 + 96) * n},
            "cpu_baseline": {"value": m / cpu_s, "unit": "points/s", "cores": ncpu, "kind": "port", "cpu_model": _cpu_model(), "seconds": cpu_s,
                             "sample": f"{m} of the encodings: square root by exponentiation + endomorphism subgroup test in C "
                                       "(oracle/msm_oracle.c orc_g1_deserialize_batch, mode 1)"}}


def _in_process_leg(pkg, co, torch, ncpu, slots: int, log_n: int, steps: int) -> dict:
    """The in-library multi-device path INTEGRATION.md binds (one context over several devices, persistent per-device host
    threads, no RCCL): `slots` device slots over the visible GPUs (round-robin; on a one-GPU box device 0 listed `slots`
    times — a rehearsal of the host plumbing, the slots then share the chip)."""
    ndev = torch.cuda.device_count()
    if slots <= 0:   # default: every visible GPU (at most 8) once; on a one-GPU box device 0 twice (plumbing rehearsal)
        slots = min(8, ndev) if ndev > 1 else 2
    ids = [k % ndev for k in range(slots)]
    n = 1 << log_n
    bases = co.gen_bases("g1", SEED_B + 7, n, ncpu)
    scalars = co.gen_scalars(SEED_S + 7, n)
    d_sc = torch.frombuffer(bytearray(scalars), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    out = {}
    with pkg.Context(ids) as c:
        c.set_bases("g1", bases, n)
        for name, fn in (("device_scalars", lambda: c.msm_device("g1", d_sc.data_ptr(), n, pkg.SCALAR_CANONICAL)),
                         ("host_scalars", lambda: c.msm("g1", None, scalars, n, pkg.SCALAR_CANONICAL))):
            fn()
            t0 = time.perf_counter()
            for _ in range(steps):
                r = fn()
            dt = (time.perf_counter() - t0) / steps
            out[name] = {"value": n / dt, "unit": "points/s", "ms_per_call": dt * 1e3,
                         "bit_exact": co.to_affine("g1", r) == co.dlog_expected("g1", scalars, SEED_B + 7, n)}
    # the same work as `slots` back-to-back single-device calls of n / slots points on device 0
    per = n // slots
    with pkg.Context([0]) as c1:
        c1.set_bases("g1", bases[:96 * per], per)
        c1.msm_device("g1", d_sc.data_ptr(), per, pkg.SCALAR_CANONICAL)
        t0 = time.perf_counter()
        for _ in range(steps):
            for _k in range(slots):
                c1.msm_device("g1", d_sc.data_ptr(), per, pkg.SCALAR_CANONICAL)
        dt = (time.perf_counter() - t0) / steps
    out["serial_single_device_calls_ms"] = dt * 1e3
    out.update({"device_slots": ids, "points": n, "plumbing_only": len(set(ids)) < len(ids),
                "note": "one mi_ctx over several device slots (mi_msm_init with a device list): contiguous shards, one persistent host "
                        "thread per slot, partial sums added in slot order; compared with the same shards as back-to-back "
                        "single-device calls"})
    return out


def _in_process_isolated(pkg, co, torch, ncpu, slots: int) -> dict:
    """The in-library multi-device leg.  With ONE visible GPU it runs here (device 0 listed twice: the plumbing every round has
    tested).  With several it sees distinct physical devices — peer access, one resident shard and one host thread per GPU — for the first
    time on whatever node runs this, so it runs in a child process (started, not exec'ed): a fault there costs this leg, not the line."""
    if torch.cuda.device_count() <= 1:
        return _in_process_leg(pkg, co, torch, ncpu, slots, 20, 5)
    if _under_profiler():
        return {"skipped": "under rocprofv3 this process must not start GPU child processes; run the leg unprofiled"}
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--in-process-child", "--in-process", str(slots)],
                           capture_output=True, text=True, timeout=120, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": "child process exceeded 120 s (a normal run takes under 30 s)"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"child process rc={r.returncode}", "stderr_tail": r.stderr[-600:]}
    out = json.loads(lines[-1])
    out["isolated_in_child_process"] = True
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=None,
                    help="log2 of points PER GPU.  Default: 20 at N = 1 (BASELINE config #2); at N > 1 the 2^24 points of config #3 "
                         "are split over the ranks instead (strong scaling)")
    ap.add_argument("--total-log-n", type=int, default=24, help="log2 of the TOTAL points at N > 1 when --log-n is not given")
    ap.add_argument("--group", default="g1", choices=["g1", "g2"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--precomputed", action="store_true",
                    help="opt-in mode: resident 2^(c j) P tables (mi_msm_g1_set_bases_precomputed); never the default headline")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "zero_one", "small64", "all_equal", "all_ones", "r1cs_mix"],
                    help="scalar distribution (secondary robustness figures; the headline is uniform)")
    ap.add_argument("--no-validate", action="store_true",
                    help="skip mi_msm_g1_validate_bases after the base upload: the MSMs then recode the integer scalar (an unvalidated SRS)")
    ap.add_argument("--no-secondary", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--secondary-out", default=None, metavar="FILE",
                    help="where the full records of the secondary legs go (default: bench_secondary.json beside bench.py); the stdout "
                         "line carries a one-line recap of each under `summary` and this file's name under `secondary_file`")
    ap.add_argument("--in-process", type=int, default=0, metavar="SLOTS",
                    help="N = 1 only: device slots of the in-library multi-GPU leg of the secondary set (default: every visible GPU, at most 8; "
                         "device 0 twice on a one-GPU box)")
    ap.add_argument("--in-process-child", action="store_true", help=argparse.SUPPRESS)   # runs ONLY the in-library multi-device leg (see _in_process_isolated)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-device rehearses the N>1 path on a single-GPU box")
    ap.add_argument("--share-device", action="store_true", help="all ranks use GPU 0 (rehearsal only)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the N > 1 exchange path (mi_msm_g1_allgather_fold: device_windows -> ncclAllGather -> one D2H -> fold_windows, all "
                         "inside libarkblst_amd_rccl.so) at any world size, including 1: a one-GPU box then executes the RCCL calls under a "
                         "one-rank communicator")
    args = ap.parse_args()

    # dmabuf IPC: RCCL across processes needs it on this pool (the host driver has no legacy IPC).  Set before torch / HIP / RCCL load, for
    # ranks started by an external launcher as much as for the ones _self_launch starts.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if _under_profiler():
            raise SystemExit("bench.py --gpus N under rocprofv3 must be started through the launcher form (python -m torch.distributed.run ... "
                             "bench.py): the profiler's preload has initialised the GPU in this process, which must then not start children")
        sys.exit(_self_launch(args.gpus))   # nothing below has run: this process never initialises the GPU

    import torch  # plumbing only: device buffers, synchronize, torch.distributed (RCCL)
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("BENCH_LAUNCH_PROBE") == "1":
        # launcher rehearsal for the CPU suite (tests/test_multigpu_gloo.py): the ranks meet over gloo, rank 0 prints ONE line
        dist.init_process_group("gloo")
        seen = [None] * world
        dist.all_gather_object(seen, (rank, local_rank))
        if rank == 0:
            print(json.dumps({"launch_probe": True, "n_gpus": world, "ranks": sorted(r for r, _ in seen), "argv": sys.argv[1:]}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the MSM path")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    on_gpu = args.backend == "nccl"   # collectives on device tensors (RCCL) or on host tensors (gloo rehearsal)
    exchange = world > 1 or args.force_exchange   # the N > 1 step; --force-exchange runs it under a one-rank group too
    if exchange:
        if world == 1:   # no launcher set the rendezvous up
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if on_gpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    import __graft_entry__ as ge
    from oracle import coracle as co   # input generation, checker, cpu_baseline only

    pkg = ge.load_package()
    g = args.group
    ncpu = _host_threads()
    if args.in_process_child:   # child of _in_process_isolated: this leg only, one JSON line
        print(json.dumps(_in_process_leg(pkg, co, torch, ncpu, args.in_process, 20, 5)), flush=True)
        return
    if world > 1:
        ncpu = max(1, min(ncpu, (os.cpu_count() or world) // world))   # the ranks share the node's cores
    aff = 96 if g == "g1" else 192

    # ---- workload: rank r owns a contiguous shard of the global index range
    if args.log_n is not None or world == 1:
        log_n = args.log_n if args.log_n is not None else 20
        n = 1 << log_n
        total = n * world
        scaling = "weak"
        wl = f"{g.upper()} MSM, 2^{log_n} random bases+scalars per GPU"
    else:
        total = 1 << args.total_log_n
        lo, hi = (total * rank) // world, (total * (rank + 1)) // world
        n = hi - lo
        log_n = n.bit_length() - 1 if n & (n - 1) == 0 else None
        scaling = "strong"
        wl = f"{g.upper()} MSM, 2^{args.total_log_n} random bases+scalars in total, base set sharded contiguously over {world} GPUs"
    wl += (", bases resident as precomputed 2^(c j) P tables" if args.precomputed else ", bases resident") + ", scalars in HBM"

    leg = MsmLeg(pkg, co, torch, g, n, SEED_B + 1000 * rank, SEED_S + 1000 * rank, ncpu, local_rank, dist=args.dist,
                 window_bits=args.window_bits, precomputed=args.precomputed, validate=not args.no_validate)

    jac_bytes = 144 if g == "g1" else 288
    cdev = "cuda" if on_gpu else "cpu"
    # ---- exchange step at N > 1, INSIDE the product: mi_msm_g{1,2}_allgather_fold (libarkblst_amd_rccl.so) leaves this
    # rank's per-window sums in device memory, all-gathers them with ncclAllGather over xGMI, copies the gathered block to the host ONCE
    # and folds — identical result on every rank.  torch only starts the ranks, carries the 128-byte ncclUniqueId to them and
    # brackets the timed region (barrier, max over ranks).  `--backend gloo --share-device` (several ranks on ONE GPU, where RCCL
    # refuses to form a communicator) rehearses the same steps with a host collective instead.
    xchg = {"info": None, "gather_host": None, "msm_s": 0.0, "exchange_s": 0.0, "steps": 0, "backend": "rccl (in-library)" if on_gpu else "gloo",
            "comm": None, "repeats": 0,
            "path": ("mi_msm_%s_allgather_fold: window sums in device memory -> ncclAllGather -> one D2H of the gathered block -> mi_%s_fold_windows" % (g, g))
            if on_gpu else ("rehearsal: mi_msm_%s_device_windows -> host all_gather (gloo) -> mi_%s_fold_windows" % (g, g))}
    win_dev = torch.empty(pkg.MAX_WINDOWS * jac_bytes, dtype=torch.uint8, device="cuda") if exchange else None

    def setup_exchange():
        if on_gpu:
            # The in-library exchange first.  It has never met a second physical GPU: if ANY rank cannot set it up — its own
            # communicator next to torch's, the first collective — every rank falls back to the same steps with torch.distributed's
            # all-gather on device tensors (RCCL as well), and the line says so.  The decision is collective: a rank never waits in a
            # collective the others have left (the library's deadline ends a wait on a rank that failed earlier).
            err = None
            try:
                if os.environ.get("ARKBLST_AMD_BENCH_EXCHANGE") == "torch":
                    raise RuntimeError("ARKBLST_AMD_BENCH_EXCHANGE=torch")
                uid = [pkg.rccl_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                xchg["comm"] = pkg.RcclComm(leg.ctx, uid[0], world, rank)   # collective: ncclCommInitRank
                xchg["comm"].allgather_fold(g, leg.d_scalars.data_ptr(), n, pkg.SCALAR_CANONICAL)   # untimed: ragged shards agree on a window size here
                t = xchg["comm"].timing()
                xchg["info"] = (t["window_bits"], t["num_windows"])
                xchg["bytes_per_rank"] = t["bytes_per_rank"]
            except Exception as e:   # noqa: BLE001 - whatever went wrong, the other ranks must learn of it
                err = f"{type(e).__name__}: {e}"
            flag = torch.tensor([0 if err else 1], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                return
            errs = [None] * world
            dist.all_gather_object(errs, err)
            xchg["comm"] = None
            xchg["backend"] = "rccl (torch.distributed all_gather on device tensors)"
            xchg["fallback_reason"] = next((e for e in errs if e), "unknown")
            xchg["path"] = "FALLBACK: mi_msm_%s_device_windows -> torch.distributed all_gather_into_tensor (device, RCCL) -> one D2H -> mi_%s_fold_windows" % (g, g)
            if rank == 0:
                print(f"[bench] in-library exchange unavailable ({xchg['fallback_reason']}); using torch.distributed's all-gather", file=sys.stderr, flush=True)
        # torch collective (gloo rehearsal, or the fallback above): learn the window geometry, make every rank agree on it, size the gather buffer
        info = leg.ctx.msm_device_windows(g, leg.d_scalars.data_ptr(), n, pkg.SCALAR_CANONICAL, win_dev.data_ptr())
        infos = [None] * world
        dist.all_gather_object(infos, info)
        if len(set(infos)) != 1:   # ragged shards chose different window sizes: pin the largest everywhere
            leg.ctx.set_window_bits(max(i[0] for i in infos))
            info = leg.ctx.msm_device_windows(g, leg.d_scalars.data_ptr(), n, pkg.SCALAR_CANONICAL, win_dev.data_ptr())
            dist.all_gather_object(infos, info)
            assert len(set(infos)) == 1, infos
        xchg["info"] = info
        xchg["bytes_per_rank"] = info[1] * jac_bytes
        xchg["gather_host"] = torch.empty(world * info[1] * jac_bytes, dtype=torch.uint8, device=cdev)

    def step() -> bytes:
        if not exchange:
            return leg.call()
        if on_gpu and xchg["comm"] is not None:
            out = xchg["comm"].allgather_fold(g, leg.d_scalars.data_ptr(), n, pkg.SCALAR_CANONICAL)
            t = xchg["comm"].timing_raw()
            xchg["msm_s"] += t.msm_ms * 1e-3
            xchg["exchange_s"] += t.exchange_ms * 1e-3
            xchg["repeats"] += t.repeats
            xchg["steps"] += 1
            return out
        t_a = time.perf_counter()
        info = leg.ctx.msm_device_windows(g, leg.d_scalars.data_ptr(), n, pkg.SCALAR_CANONICAL, win_dev.data_ptr())
        t_b = time.perf_counter()
        nb = info[1] * jac_bytes
        dist.all_gather_into_tensor(xchg["gather_host"], win_dev[:nb] if on_gpu else win_dev[:nb].cpu())
        out = pkg.fold_windows(g, xchg["gather_host"].cpu().numpy(), world, info[1], *info)
        t_c = time.perf_counter()
        xchg["msm_s"] += t_b - t_a
        xchg["exchange_s"] += t_c - t_b
        xchg["steps"] += 1
        return out

    def fence():
        if exchange:
            dist.barrier()
        torch.cuda.synchronize()

    if exchange:
        setup_exchange()
    for _ in range(args.warmup):
        step()
    prof_acc = []
    xchg.update(msm_s=0.0, exchange_s=0.0, steps=0, repeats=0)
    fence()
    t0 = time.perf_counter()
    result = b""
    for _ in range(args.steps):
        result = step()
        prof_acc.append(leg.ctx.profile_raw())   # the C struct only: the dicts are built after the timed region
    fence()
    elapsed = time.perf_counter() - t0
    prof_acc = [pkg.profile_dict(p) for p in prof_acc]
    if exchange:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # how a step splits: every rank's local part (up to its window sums in device memory) — the slowest rank sets the pace, the
        # others wait for it inside the all-gather: wait_ms = slowest local part - rank 0's; what remains of rank 0's exchange time
        # is the collective proper + the one D2H + the host fold.  No barrier inside the timed steps.
        ks = max(1, xchg["steps"])
        mine = torch.tensor([xchg["msm_s"] / ks * 1e3], dtype=torch.float64, device=cdev)
        allm = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        xchg["rank_msm_ms"] = [float(t.item()) for t in allm]

    breakdown = leg.phase_breakdown(3 if n <= (1 << 22) else 1)   # right after the timed region (warm clocks): every phase's events

    # ---- parity: closed form over ALL ranks' inputs
    mine_expected = leg.expected_affine()          # affine bytes of this rank's shard
    if exchange:
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

    secondary = {}
    if world == 1 and not args.no_secondary and rank == 0:
        # ---- the same K steps issued by TWO host threads on the same context (the trait method is re-entrant and arkworks
        # calls it from rayon workers): a context has two lanes.  Not the headline value.
        import threading
        res2 = [b""] * args.steps
        def worker(t):
            for k in range(t, args.steps, 2):
                res2[k] = leg.call()
        worker(1)   # first use of the second lane allocates its scratch
        fence()
        t1 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
        for x in th: x.start()
        for x in th: x.join()
        fence()
        e2 = time.perf_counter() - t1
        secondary["two_host_threads"] = {"value": n * args.steps / e2, "unit": "points/s", "ms_per_step": e2 / args.steps * 1e3,
                                         "same_result": len({co.to_affine(g, r) for r in res2 + [result]}) == 1,
                                         "note": "the same K steps issued by two host threads on ONE context (two lanes, shared resident bases)"}
        # ---- end-to-end call shapes (SURVEY 8(d) "timing scope"): scalars from host memory per call with resident bases, and
        # the reference driver's shape — bases AND scalars uploaded on every call (src/gpu.rs:149-150)
        def best_of(fn, reps=3):
            fn()
            b = 1e30
            for _ in range(reps):
                t1 = time.perf_counter(); r = fn(); b = min(b, time.perf_counter() - t1)
            assert co.to_affine(g, r) == co.to_affine(g, result)
            return b * 1e3
        if not args.precomputed:
            shapes = {
                "resident_bases_host_scalars_ms": best_of(lambda: leg.ctx.msm(g, None, leg.scalars, n, pkg.SCALAR_CANONICAL)),
                "host_bases_host_scalars_ms": best_of(lambda: leg.ctx.msm(g, leg.bases, leg.scalars, n, pkg.SCALAR_CANONICAL))}
            # the trait's own shape — host slices on every call, no handle (src/g1.rs:604) — with the base-set cache the shim of
            # INTEGRATION.md §2 turns on: the first call with a slice converts and keeps it, later ones find it by fingerprint
            leg.ctx.set_base_cache(2)
            shapes["host_bases_cached_ms"] = best_of(lambda: leg.ctx.msm(g, leg.bases, leg.scalars, n, pkg.SCALAR_CANONICAL))
            shapes["base_cache"] = leg.ctx.base_cache_stats()
            leg.ctx.set_base_cache(0)
            shapes["note"] = ("per call incl. H2D of the scalars (and bases) from pageable host memory; the headline keeps both in HBM.  A Rust caller "
                              "of the trait method gets host_bases_cached_ms from the second call with the same base slice on (cache on in the shim), "
                              "host_bases_host_scalars_ms with ARKBLST_AMD_BASE_CACHE=0")
            secondary["call_shapes"] = shapes

    # ---- CPU baseline (rank 0, N = 1): SURVEY 8(d): one warm-up + median of >= 5 runs (3 above 2^20), CPU model and core count in
    # the result, plus a single-thread figure (on a 2^16-point prefix) for scaling
    cpu_baseline = None
    if not args.no_cpu_baseline and world == 1 and rank == 0:
        runs = 5 if n <= (1 << 20) else (3 if n <= (1 << 22) else 1)
        if n <= (1 << 22):
            co.msm(g, leg.bases, leg.scalars, n, 0, ncpu)
        times = []
        for _ in range(runs):
            t0 = time.perf_counter()
            cpu = co.msm(g, leg.bases, leg.scalars, n, 0, ncpu)
            times.append(time.perf_counter() - t0)
        med = statistics.median(times)
        assert co.to_affine(g, cpu) == leg.expected_affine()
        n1 = min(n, 1 << 20)   # one thread on (up to) the 2^20-point workload: ~5 s
        t0 = time.perf_counter()
        co.msm(g, leg.bases[:aff * n1], leg.scalars[:32 * n1], n1, 0, 1)
        t_single = time.perf_counter() - t0
        cpu_baseline = {"value": n / med, "unit": "points/s", "cores": ncpu, "kind": "port", "cpu_model": _cpu_model(),
                        "sample": f"full workload of one GPU ({n} points), median of {runs} runs after a warm-up, blst-style Pippenger "
                                  "restatement in C with a mulx/adcx/adox Montgomery multiplication (oracle/msm_oracle.c, field.h), not blst's assembly",
                        "seconds": med, "best_seconds": min(times),
                        "single_thread": {"value": n1 / t_single, "unit": "points/s", "points": n1}}

    headline_n, gen_s, validated, validate_s = n, leg.gen_s, leg.validated, leg.validate_s
    p0 = prof_acc[-1]
    acc_ms = sum(p["accumulate_ms"] for p in prof_acc) / len(prof_acc)
    if xchg.get("comm") is not None:
        if world > 1:
            dist.barrier()       # ncclCommDestroy is collective-ish: nobody tears down while a peer is still inside a step
        xchg["comm"].close()     # before the context it sits on
    leg.close()
    del leg

    if world == 1 and not args.no_secondary and rank == 0:
        # ---- the other single-GPU configs of BASELINE.json, each on a fresh context (never let one break the headline line)
        def guarded(name, fn):
            try:
                secondary[name] = fn()
            except Exception as e:
                secondary[name] = {"error": repr(e)}
        if g == "g1" and log_n == 20 and not args.precomputed and args.dist == "uniform":
            guarded("g1_2p16", lambda: _secondary_msm(pkg, co, torch, "g1", 16, 16, ncpu, local_rank, 50))   # config #1's size (the reference's CPU-runnable case)
            guarded("g1_2p24", lambda: _secondary_msm(pkg, co, torch, "g1", 24, 24, ncpu, local_rank, 3, cpu_sample_log_n=22))
            guarded("g2_2p20", lambda: _secondary_msm(pkg, co, torch, "g2", 20, 4, ncpu, local_rank, 5))
            guarded("g1_2p20_precomputed_tables", lambda: _secondary_msm(pkg, co, torch, "g1", 20, 0, ncpu, local_rank, 10, precomputed=True))
        guarded("pairing_2p16", lambda: _pairing_leg(pkg, co, ncpu, local_rank))
        if g == "g1" and log_n == 20 and not args.precomputed and args.dist == "uniform":
            guarded("normalize_2p20", lambda: _normalize_leg(pkg, co, ncpu, local_rank))
            guarded("deserialize_2p20", lambda: _deserialize_leg(pkg, co, ncpu, local_rank))
            guarded("normalize_g2_2p20", lambda: _normalize_leg(pkg, co, ncpu, local_rank, "g2"))
            guarded("deserialize_g2_2p18", lambda: _deserialize_g2_leg(pkg, co, ncpu, local_rank))
        guarded("in_process_multi_device", lambda: _in_process_isolated(pkg, co, torch, ncpu, args.in_process))

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = total / (elapsed / args.steps)
        full = _rooflines(g, headline_n, log_n, acc_ms, p0["num_windows"], args.precomputed, _clock_of(prof_acc))
        roof = full["roofline"]
        # whole-step fraction of the same roof: the model's multiply-adds of ALL ranks' points over the step time (sort, bucket reduction,
        # combine, host fold and exchange included) against the peak of the GPUs used; `roofline.frac` is the dominant kernel alone
        step_frac = roof["model_mads_per_point"] * total / (ms_step * 1e-3) / 1e12 / (MAD_PEAK_T * world)
        out = {
            "metric": f"{g.upper()} MSM points/sec",
            "value": value,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "bit_exact": bit_exact,
            "config": {"workload": wl, "points_per_gpu": headline_n, "total_points": total, "window_bits": p0["window_bits"],
                       "num_windows": p0["num_windows"], "parallelism": f"base-set sharded x{world}", "scalar_dist": args.dist,
                       "precomputed_tables": bool(args.precomputed), "bases_validated": validated, "field_repr": "14 x 28-bit limbs in u32 (v_mad_u64_u32)"},
            "roofline": _compact(roof, _ROOF_KEYS),
            "hbm_roofline": _compact(full["hbm_roofline"], _HBM_KEYS),
            "step_frac": step_frac,
            "phases_ms": {k: round(v, 4) for k, v in _phases(prof_acc, breakdown).items()},
        }
        detail = {"headline_roofline": roof, "headline_hbm_roofline": full["hbm_roofline"], "input_gen_s": gen_s, "validate_bases_s": validate_s}
        if exchange:
            exp_ms, exp_src = _expected_ms(g, log_n) if log_n is not None else (None, None)
            out["config"]["expected_ms_per_rank"] = exp_ms
            ks = max(1, xchg["steps"])
            rm = xchg["rank_msm_ms"]
            out["msm_ms"] = rm[0]                                             # rank 0: the local pipeline up to its window sums in device memory
            out["wait_ms"] = max(0.0, max(rm) - rm[0])                        # rank 0 inside the all-gather until the slowest rank arrives (load imbalance)
            out["exchange_ms"] = max(0.0, xchg["exchange_s"] / ks * 1e3 - out["wait_ms"])   # header + all-gather + one D2H + host fold
            out["exchange"] = {"backend": xchg["backend"], "world_size": world, "bytes_per_rank": xchg["bytes_per_rank"],
                               "windows": xchg["info"][1], "window_bits": xchg["info"][0], "path": xchg["path"],
                               "rank_msm_ms": [round(x, 4) for x in rm], "exchange_incl_wait_ms": xchg["exchange_s"] / ks * 1e3,
                               "window_size_repeats_in_timed_steps": xchg["repeats"]}
            if xchg.get("fallback_reason"):
                out["exchange"]["fallback_reason"] = xchg["fallback_reason"][:300]
            detail["exchange"] = {"expected_ms_source": exp_src}
        if cpu_baseline:
            out["cpu_baseline"] = cpu_baseline
        # compact recap: the north-star figures of every leg (the full records of the legs go to the sidecar file)
        def _brief(d):
            b = {"value": d.get("value"), "ms": d.get("ms_per_step", d.get("ms", d.get("call_ms_host_buffers", d.get("kernel_ms", d.get("kernels_ms"))))),
                 "bit_exact": d.get("bit_exact")}
            if "call_ms_host_buffers" in d:   # rows (f): value and ms are the CALL (host slices in and out); the kernels' time beside it
                b["kernel_ms"] = d.get("kernel_ms", d.get("kernels_ms"))
            if "roofline" in d and d["roofline"].get("bound") == "valu_int_mad":
                b["valu_frac"] = round(d["roofline"]["frac"], 3)
            if "step_frac" in d:
                b["step_frac"] = round(d["step_frac"], 3)
            if "kernel_ms" in d and "roofline" in d and d is not out:
                b["kernel_ms"] = d["kernel_ms"]
                b["clock_ghz"] = d["roofline"].get("shader_clock_ghz")
            if d.get("cpu_baseline") and d is not out:
                b["cpu"] = d["cpu_baseline"]["value"]
            return {k: (float("%.4g" % v) if isinstance(v, float) else v) for k, v in b.items()}
        summary = {f"{g}_2p{log_n}" if log_n is not None else f"{g}_{headline_n}": _brief(out)}
        for k, v in secondary.items():
            if isinstance(v, dict) and "error" in v:
                summary[k] = {"error": str(v["error"])[:120]}
            elif isinstance(v, dict) and "value" in v:
                summary[k] = _brief(v)
        if "call_shapes" in secondary:
            summary["call_shapes_ms"] = {k: round(v, 3) for k, v in secondary["call_shapes"].items() if isinstance(v, float)}
        out["summary"] = summary
        # ---- the sidecar: every secondary leg in full (rooflines with their notes, cpu baselines, phase tables).  The line below
        # stays a few KB: a record the driver can parse whole (round 4's 40 KB line was not)
        if secondary or args.secondary_out:
            path = args.secondary_out or os.path.join(ROOT, "bench_secondary.json")
            try:
                with open(path, "w") as f:
                    json.dump({"headline": out, "detail": detail, "secondary": secondary}, f, indent=1)
                out["secondary_file"] = os.path.relpath(path, ROOT)
            except OSError as e:
                out["secondary_file"] = None
                out["secondary_file_error"] = repr(e)[:200]
        line = json.dumps(out)
        if len(line) > MAX_LINE_BYTES:   # never again a line the driver cannot parse: drop the optional blocks, largest first
            for k in ("phases_ms", "summary", "exchange", "hbm_roofline"):
                out.pop(k, None)
                line = json.dumps(out)
                if len(line) <= MAX_LINE_BYTES:
                    break
        sys.stdout.flush()
        print(line, flush=True)
    if exchange:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
