#!/usr/bin/env python3
"""Every loop (backward-branch span) of one kernel of the shipped library with its instruction mix, priced with the measured
per-instruction issue costs of tools/kernel_resources.py — the per-trip accounting behind DESIGN.md's pairing section.

    python tools/loop_mix.py <kernel-name-substring> [lib.so] [--top N]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr


def loops_of(needle, lib=None):
    lib = lib or os.path.join(kr.ROOT, "ark-blst_amd", "lib", "libarkblst_amd.so")
    with tempfile.TemporaryDirectory() as tmp:
        for elf in kr.code_objects(lib, tmp):
            txt = subprocess.run([os.path.join(kr.LLVM, "llvm-objdump"), "-d", elf], capture_output=True, text=True, check=True).stdout
            sym, base, ins, found = None, 0, [], None
            for line in txt.splitlines():
                m = re.match(r"^([0-9a-f]{16}) <(.+)>:$", line)
                if m:
                    if found:
                        break
                    sym, base, ins = m.group(2), int(m.group(1), 16), []
                    found = sym if needle in kr.demangle([sym])[sym] else None
                    continue
                m = re.match(r"^\s+(\S+).*// ([0-9A-F]{12}):", line)
                if m and found:
                    tgt = None
                    if m.group(1).startswith(("s_cbranch", "s_branch")):
                        t = re.search(r"<[^>]*\+0x([0-9a-f]+)>", line)
                        tgt = int(t.group(1), 16) if t else 0
                    ins.append((int(m.group(2), 16) - base, m.group(1), tgt))
            if found:
                out = []
                for off, mn, tgt in ins:
                    if tgt is not None and tgt < off:
                        out.append((tgt, off, [i[1] for i in ins if tgt <= i[0] <= off]))
                return kr.demangle([found])[found], len(ins), sorted(out)
    return None, 0, []


def price(body):
    c = collections.Counter(body)
    return sum(kr.COST.get(k, kr.DEFAULT_VALU if k.startswith("v_") else kr.DEFAULT_OTHER) * v for k, v in c.items()), c


if __name__ == "__main__":
    argv = sys.argv[1:]
    top = 14
    if "--top" in argv:
        i = argv.index("--top")
        top = int(argv[i + 1])
        del argv[i:i + 2]
    args = argv
    name, total, loops = loops_of(args[0], args[1] if len(args) > 1 else None)
    print(f"{name}: {total} instructions, {len(loops)} loops")
    for lo, hi, body in loops:
        cost, c = price(body)
        mads = c["v_mad_u64_u32"]
        print(f"  loop {lo:#x}..{hi:#x}: {len(body)} instructions, {mads} v_mad_u64_u32 ({mads / 196:.1f} x 196), {len(body) - mads} others, "
              f"priced {cost:.0f} (2.4-GHz pseudo-cycles)")
        print("     ", ", ".join(f"{k} {v}" for k, v in c.most_common(top)))
