#!/usr/bin/env python3
"""profiles/r05_instruction_costs.json: what the hot loops of the shipped code objects cost a SIMD if every instruction issues at its
measured rate (tools/kernel_resources.py COST: tools/ubench_carry.hip / ubench_mad_banks.hip, two waves per SIMD).  Those figures
were taken as time x 2.4 GHz; the kernels run at the shader clock the PMC passes measure (GRBM_GUI_ACTIVE / duration, ~2.04 GHz
under this load — and the micro-benchmarks, pure multiply-add streams, are under the same power limit), so every cost is rescaled by
clock / 2.4 into REAL cycles.  bench.py reads the file for `instruction_cost_sum_cycles` / `frac_of_instruction_cost_bound`.

    python tools/instruction_costs.py [shader_clock_ghz]     (default: from the newest profiles/*_pmc_summary.json that has one)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr
from loop_mix import loops_of, price

ROOT = kr.ROOT


def clock_from_profiles(key):
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted((f for f in os.listdir(pdir) if f.endswith("_pmc_summary.json")), reverse=True):
        for k, v in json.load(open(os.path.join(pdir, f))).get("kernels", {}).items():
            if key in k and "shader_clock_ghz" in v:
                return v["shader_clock_ghz"], "profiles/" + f
    return None, None


def biggest(needle, min_mads=0):
    name, total, loops = loops_of(needle)
    out = []
    for lo, hi, body in loops:
        cost, c = price(body)
        out.append({"span": [lo, hi], "instructions": len(body), "v_mad_u64_u32": c["v_mad_u64_u32"], "pseudo_cycles": cost})
    return name, [x for x in out if x["v_mad_u64_u32"] >= min_mads]


def main():
    out = {"note": __doc__.split("\n\n")[0]}
    clk = float(sys.argv[1]) if len(sys.argv) > 1 else None
    for g, needle, per_trip in (("g1", "k_accumulate<msmk::G1C>", 64), ("g2", "k_accumulate_g2_coop", 32)):
        c, src = (clk, "command line") if clk else clock_from_profiles(needle.split("<")[0] if g == "g2" else "k_accumulate<msmk::G1C>")
        if c is None:
            c, src = 2.04, "default 2.04 GHz (no PMC summary with a shader clock found)"
        name, loops = biggest(needle, 1000)
        hot = max(loops, key=lambda x: x["v_mad_u64_u32"])
        scale = 64 / per_trip   # trips per wave-wide addition (the G2 kernel holds 32 additions per wave: a lane pair per work item)
        out[g] = {"kernel": name.split("(")[0], "instructions_per_trip": hot["instructions"], "v_mad_u64_u32_per_trip": hot["v_mad_u64_u32"],
                  "additions_per_wave_trip": per_trip, "pseudo_cycles_per_trip_at_2p4ghz": hot["pseudo_cycles"], "shader_clock_ghz": c,
                  "cycles": hot["pseudo_cycles"] * scale * c / 2.4,
                  "source": f"tools/instruction_costs.py: hot loop of the shipped code object priced with tools/kernel_resources.py COST, "
                            f"rescaled to the measured shader clock ({src}); per 64 additions of one wave"}
    # pairing: the accumulate kernel's two inner loops (one line product per trip; one squaring term per trip) and the line kernel's loop
    c, src = (clk, "command line") if clk else clock_from_profiles("k_miller_accumulate")
    if c is None:
        c, src = 2.0, "default 2.0 GHz"
    _, acc = biggest("k_miller_accumulate", 500)
    line = [x for x in acc if 1900 <= x["v_mad_u64_u32"] <= 2400]
    sqt = [x for x in acc if 500 <= x["v_mad_u64_u32"] <= 700]
    _, lines = biggest("k_miller_lines2", 5000)
    if line and sqt and lines:
        lp, st, lk = line[0], sqt[0], lines[0]
        sq_total = 4 * st["pseudo_cycles"] + 2 * 196 * 4.8 + 300   # four term trips + the reduction of two components + stores
        n, m, groups = 1 << 16, 7, 10
        acc_cycles = (63 * sq_total + 68 * m * lp["pseudo_cycles"]) * c / 2.4          # one wave, it has a SIMD to itself
        lines_cycles = 2 * 68 * lk["pseudo_cycles"] / 68 * (63 + 5 * 1.6) * c / 2.4     # two waves share a SIMD; an addition step ~1.6 doubling steps
        out["pairing"] = {"shader_clock_ghz": c, "shader_clock_source": src,
                          "k_miller_accumulate": {"line_product_trip": lp, "squaring_term_trip": st, "pairs_per_wave": m * groups,
                                                  "instruction_cost_ms_2p16_pairs": acc_cycles / (c * 1e9) * 1e3,
                                                  "trip": "one iteration of the pair loop = f <- f * line for the ten accumulators of a wave (six lanes each)"},
                          "k_miller_lines2": {"step_loop": lk, "instruction_cost_ms_2p16_pairs": lines_cycles / (c * 1e9) * 1e3,
                                              "trip": "one iteration of the bit loop = one doubling step (plus, statically, the addition step body "
                                                      "taken 5 times in 63) for the 32 pairs of a wave (two lanes each)"}}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r05_instruction_costs.json"), "w"), indent=1)
    print(json.dumps({k: (v if k == "note" else {a: b for a, b in v.items() if not isinstance(b, dict)}) for k, v in out.items()}, indent=1)[:1500])


if __name__ == "__main__":
    main()
