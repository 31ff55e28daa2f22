#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of the SHIPPED library, read from the gfx950 code objects inside it.

    python tools/kernel_resources.py [path/to/lib.so] [--json]

The .so carries one clang offload bundle per translation unit in its .hip_fatbin section; each bundle holds one gfx950 ELF
whose NT_AMDGPU_METADATA note lists, per kernel, .vgpr_count / .agpr_count / .vgpr_spill_count / .sgpr_spill_count /
.private_segment_fixed_size (scratch bytes per lane) / .group_segment_fixed_size (LDS bytes per workgroup).  This is what
`llvm-readelf --notes` prints; tests/test_cabi.py fails the CPU suite when a hot kernel reports spills, so that DESIGN.md
cannot drift from the binary.
"""
from __future__ import annotations

import json
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _fatbin(lib: str, tmp: str) -> bytes:
    out = os.path.join(tmp, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={out}", lib, os.path.join(tmp, "copy.so")])
    with open(out, "rb") as f:
        return f.read()


def code_objects(lib: str, tmp: str) -> list[str]:
    """Write every gfx950 ELF of the library into tmp; returns their paths."""
    data = _fatbin(lib, tmp)
    paths = []
    for m in re.finditer(re.escape(MAGIC), data):
        base = m.start()
        (nent,) = struct.unpack_from("<Q", data, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(nent):
            off, size, tlen = struct.unpack_from("<QQQ", data, pos)
            triple = data[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if "gfx950" in triple and size:
                p = os.path.join(tmp, f"co{len(paths)}.elf")
                with open(p, "wb") as f:
                    f.write(data[base + off:base + off + size])
                paths.append(p)
    return paths


_KEYS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size",
         ".group_segment_fixed_size", ".max_flat_workgroup_size")


def kernels_of(elf: str) -> dict[str, dict[str, int]]:
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], capture_output=True, text=True, check=True).stdout
    out: dict[str, dict[str, int]] = {}
    cur: dict[str, int] | None = None
    for line in txt.splitlines():
        s = line.strip()
        if s.startswith("- .agpr_count") or s.startswith("- .args"):
            cur = {}
        if cur is None:
            continue
        m = re.match(r"-?\s*(\.[a-z_]+):\s+(\S+)", s)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k in _KEYS:
            cur[k[1:]] = int(v)
        elif k == ".name":
            out[v.strip("'\"")] = cur
    return out


def demangle(names: list[str]) -> dict[str, str]:
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, r.stdout.splitlines()))


def resources(lib: str | None = None) -> dict[str, dict[str, int]]:
    lib = lib or os.path.join(ROOT, "ark-blst_amd", "lib", "libarkblst_amd.so")
    res: dict[str, dict[str, int]] = {}
    with tempfile.TemporaryDirectory() as tmp:
        for elf in code_objects(lib, tmp):
            ks = kernels_of(elf)
            names = demangle(list(ks))
            for raw, v in ks.items():
                res[names[raw]] = v
    return res


# kernels whose inner loops must run without register spills (DESIGN_HISTORY.md §5): substrings of the demangled name
HOT = ("msmk::k_accumulate", "msmk::k_reduce_", "msmk::k_combine", "msmk::k_miller_", "msmk::k_fp12_prod")


def allocated_vgprs(lib: str | None = None) -> dict[str, int]:
    """demangled kernel name -> registers per lane the hardware ALLOCATES for a wave (granulated_workitem_vgpr_count of the kernel descriptor,
    granule 8 on gfx950).  Not the same as the notes' vgpr_count: with a large static LDS array the compiler pads the allocation up to the most its
    LDS-limited occupancy allows (round 6: 176 allocated for 52 used), which decides whether a workgroup fits BESIDE another kernel."""
    import struct
    lib = lib or os.path.join(ROOT, "ark-blst_amd", "lib", "libarkblst_amd.so")
    out: dict[str, int] = {}
    with tempfile.TemporaryDirectory() as tmp:
        for elf in code_objects(lib, tmp):
            syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", elf], capture_output=True, text=True, check=True).stdout
            secs = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-S", "--wide", elf], capture_output=True, text=True, check=True).stdout
            ro = None
            for line in secs.splitlines():
                f = line.split()
                if ".rodata" in f:
                    i = f.index(".rodata")
                    ro = (int(f[i + 2], 16), int(f[i + 3], 16))
            if ro is None:
                continue
            data = open(elf, "rb").read()
            kds = []
            for line in syms.splitlines():
                f = line.split()
                if len(f) >= 8 and f[7].endswith(".kd"):
                    kds.append((f[7][:-3], int(f[1], 16)))
            names = demangle([k for k, _ in kds])
            for raw, addr in kds:
                off = addr - ro[0] + ro[1]
                rsrc1 = struct.unpack_from("<I", data, off + 48)[0]
                out[names[raw]] = ((rsrc1 & 0x3F) + 1) * 8
    return out


def is_hot(name: str) -> bool:
    return any(h in name for h in HOT)


def scratch_in_hot_loops(lib: str | None = None) -> dict[str, int]:
    """Scratch loads / stores INSIDE the arithmetic loops of the hot kernels, read off the disassembly of the shipped code objects.
    A loop = the span of a backward branch; "arithmetic" = it holds at least 500 v_mad_u64_u32 (an inlined multiplier: the cold
    loops call out-of-line bodies and pass their operands through scratch legitimately).  The code-object metadata cannot see
    this: a loop variable whose address escapes is kept in scratch and written back every iteration with vgpr_spill_count = 0
    (round 3: 7.9 GB of scratch writes per launch of the G2 accumulate kernel until the tail took a copy)."""
    lib = lib or os.path.join(ROOT, "ark-blst_amd", "lib", "libarkblst_amd.so")
    out: dict[str, int] = {}
    with tempfile.TemporaryDirectory() as tmp:
        for elf in code_objects(lib, tmp):
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", elf], capture_output=True, text=True, check=True).stdout
            sym, base, ins = None, 0, []   # ins: (offset, mnemonic, branch target offset or None)
            blocks = []
            for line in txt.splitlines():
                m = re.match(r"^([0-9a-f]{16}) <(.+)>:$", line)
                if m:
                    if sym is not None:
                        blocks.append((sym, ins))
                    sym, base, ins = m.group(2), int(m.group(1), 16), []
                    continue
                m = re.match(r"^\s+(\S+).*// ([0-9A-F]{12}):", line)
                if m and sym is not None:
                    tgt = None
                    if m.group(1).startswith(("s_cbranch", "s_branch")):
                        t = re.search(r"<[^>]*\+0x([0-9a-f]+)>", line)
                        tgt = int(t.group(1), 16) if t else 0
                    ins.append((int(m.group(2), 16) - base, m.group(1), tgt))
            if sym is not None:
                blocks.append((sym, ins))
            names = demangle([b[0] for b in blocks])
            for raw, ins in blocks:
                name = names[raw]
                if not is_hot(name):
                    continue
                bad = 0
                for off, mn, tgt in ins:
                    if tgt is None or tgt >= off:
                        continue
                    body = [i for i in ins if tgt <= i[0] <= off]
                    if sum(1 for i in body if i[1] == "v_mad_u64_u32") >= 500:
                        bad = max(bad, sum(1 for i in body if i[1].startswith("scratch_")))
                out[name] = bad
    return out


def kernels_with_calls(lib: str | None = None) -> dict[str, bool]:
    """demangled kernel name -> its own body contains a call (s_swappc_b64), read off the disassembly of the shipped code objects.
    Together with .agpr_count (which covers the kernel's whole call graph: the assembler takes the maximum over the callees) this is
    the invariant of csrc/Makefile's compiler-issue note: a kernel with calls uses no AGPRs."""
    lib = lib or os.path.join(ROOT, "ark-blst_amd", "lib", "libarkblst_amd.so")
    out: dict[str, bool] = {}
    with tempfile.TemporaryDirectory() as tmp:
        for elf in code_objects(lib, tmp):
            kernels = set(kernels_of(elf))
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", elf], capture_output=True, text=True, check=True).stdout
            sym, found = None, {}
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]{16} <(.+)>:$", line)
                if m:
                    sym = m.group(1)
                    found.setdefault(sym, False)
                elif sym is not None and "s_swappc_b64" in line:
                    found[sym] = True
            names = demangle([k for k in found if k in kernels])
            for raw, name in names.items():
                out[name] = found[raw]
    return out


# cycles per wave-instruction at two waves per SIMD (profiles/r03_ubench_carry.txt, r03_ubench_mad_banks.txt); VOP3 encodings ~4.4-5.0, VOP1/2 ~2.6
COST = {"v_mad_u64_u32": 4.8, "v_mul_lo_u32": 4.4, "v_lshrrev_b64": 4.6, "v_lshl_add_u64": 5.05, "v_ashrrev_i64": 4.5, "v_alignbit_b32": 4.6,
        "v_lshl_add_u32": 4.4, "v_add3_u32": 4.4, "v_or3_b32": 4.4, "v_lshl_or_b32": 4.4, "v_bitop3_b32": 4.4, "v_mov_b64_e32": 4.4,
        "v_lshlrev_b64": 4.6, "v_cndmask_b32_e64": 4.4, "v_cmp_eq_u32_e64": 4.4, "v_cmp_lt_u32_e64": 4.4, "v_mov_b32_dpp": 2.7}
DEFAULT_VALU, DEFAULT_OTHER = 2.6, 1.0


def instruction_mix(lib: str | None, needle: str):
    """Histogram of the largest arithmetic loop (most v_mad_u64_u32 inside one backward-branch span) of the kernel whose demangled
    name contains `needle`, priced with COST: what one trip of the loop costs a SIMD when it issues at the measured instruction rates."""
    lib = lib or os.path.join(ROOT, "ark-blst_amd", "lib", "libarkblst_amd.so")
    with tempfile.TemporaryDirectory() as tmp:
        for elf in code_objects(lib, tmp):
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", elf], capture_output=True, text=True, check=True).stdout
            sym, base, ins, blocks = None, 0, [], []
            for line in txt.splitlines():
                m = re.match(r"^([0-9a-f]{16}) <(.+)>:$", line)
                if m:
                    if sym is not None:
                        blocks.append((sym, ins))
                    sym, base, ins = m.group(2), int(m.group(1), 16), []
                    continue
                m = re.match(r"^\s+(\S+).*// ([0-9A-F]{12}):", line)
                if m and sym is not None:
                    tgt = None
                    if m.group(1).startswith(("s_cbranch", "s_branch")):
                        t = re.search(r"<[^>]*\+0x([0-9a-f]+)>", line)
                        tgt = int(t.group(1), 16) if t else 0
                    ins.append((int(m.group(2), 16) - base, m.group(1), tgt))
            if sym is not None:
                blocks.append((sym, ins))
            names = demangle([b[0] for b in blocks])
            for raw, ins in blocks:
                if needle not in names[raw] or "k_" not in names[raw]:
                    continue
                best = None
                for off, mn, tgt in ins:
                    if tgt is None or tgt >= off:
                        continue
                    body = [i[1] for i in ins if tgt <= i[0] <= off]
                    mads = sum(1 for x in body if x == "v_mad_u64_u32")
                    if best is None or mads > best[0]:
                        best = (mads, body)
                if best:
                    return names[raw], best[1]
    return None, []


def main() -> None:
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    res = resources(args[0] if args else None)
    if "--json" in sys.argv:
        print(json.dumps(res, indent=1, sort_keys=True))
        return
    mix = [a for a in sys.argv if a.startswith("--mix=")]
    if mix:
        import collections
        name, body = instruction_mix(args[0] if args else None, mix[0].split("=", 1)[1])
        if not body:
            print("no such kernel / loop")
            return
        hist = collections.Counter(body)
        total = 0.0
        print(f"largest arithmetic loop of {name[:100]}: {len(body)} instructions (a rarely taken branch body inside the span is counted too)")
        for mn, c in hist.most_common():
            cost = COST.get(mn, DEFAULT_VALU if mn.startswith("v_") else DEFAULT_OTHER)
            total += c * cost
            print(f"  {c:6d}  {mn:24s} x {cost:4.2f} = {c * cost:9.0f} cycles")
        print(f"  sum of measured instruction costs (two waves per SIMD): {total:.0f} cycles per trip and SIMD")
        return
    if "--loops" in sys.argv:
        for k, v in sorted(scratch_in_hot_loops(args[0] if args else None).items()):
            print(f"{v:5d} scratch instructions inside arithmetic loops   {k[:110]}")
        return
    print(f"{'kernel':100s} vgpr agpr  spill scratch    lds")
    for name in sorted(res):
        v = res[name]
        hot = "*" if is_hot(name) else " "
        print(f"{hot}{name[:99]:99s} {v.get('vgpr_count', 0):4d} {v.get('agpr_count', 0):4d} {v.get('vgpr_spill_count', 0):6d} "
              f"{v.get('private_segment_fixed_size', 0):7d} {v.get('group_segment_fixed_size', 0):6d}")


if __name__ == "__main__":
    main()
