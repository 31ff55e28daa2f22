#!/usr/bin/env python3
"""Per loop of a kernel (or device function) of the shipped library: instructions, v_mad_u64_u32, scratch loads / stores and calls, read off the
disassembly.  A register-resident hot loop shows scratch 0; the rows-(f) kernels of rounds 2-5 did not (profiles/r06_rows_f_*_pmc_summary.json).
    python tools/loop_scratch.py <substring of the demangled name> [lib]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr
needle = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(kr.ROOT, "ark-blst_amd", "lib", "libarkblst_amd.so")
with tempfile.TemporaryDirectory() as tmp:
    seen = set()
    for elf in kr.code_objects(lib, tmp):
        txt = subprocess.run([os.path.join(kr.LLVM, "llvm-objdump"), "-d", elf], capture_output=True, text=True, check=True).stdout
        sym, blocks = None, {}
        for line in txt.splitlines():
            m = re.match(r"^([0-9a-f]{16}) <(.+)>:$", line)
            if m:
                sym, base = m.group(2), int(m.group(1), 16)
                blocks[sym] = []
                continue
            m = re.match(r"^\s+(\S+).*// ([0-9A-F]{12}):", line)
            if m and sym:
                tgt = None
                if m.group(1).startswith(("s_cbranch", "s_branch")):
                    t = re.search(r"<[^>]*\+0x([0-9a-f]+)>", line)
                    tgt = int(t.group(1), 16) if t else 0
                blocks[sym].append((int(m.group(2), 16) - base, m.group(1), tgt))
        names = kr.demangle(list(blocks))
        for raw, ins in blocks.items():
            n = names[raw]
            if needle not in n or n in seen:
                continue
            seen.add(n)
            cnt = lambda body, f: sum(1 for i in body if f(i[1]))
            print(f"{n[:120]}\n   whole body: {len(ins)} instructions, scratch {cnt(ins, lambda m: m.startswith('scratch_'))}, calls {cnt(ins, lambda m: m == 's_swappc_b64')}")
            for off, mn, tgt in ins:
                if tgt is not None and tgt < off:
                    body = [i for i in ins if tgt <= i[0] <= off]
                    print(f"   loop {tgt:#x}..{off:#x}: {len(body)} instructions, {cnt(body, lambda m: m == 'v_mad_u64_u32')} v_mad_u64_u32, "
                          f"scratch {cnt(body, lambda m: m.startswith('scratch_'))}, calls {cnt(body, lambda m: m == 's_swappc_b64')}")
