#!/usr/bin/env python3
"""Static check of a gfx950 assembly listing (hipcc -S --cuda-device-only) against the registers its callees really write.

For every `s_swappc_b64` whose target is a function of the same listing: every register the callee (transitively) writes and that
the caller READS after the call before writing it again (linear scan to the end of the function; control flow is ignored, so a hit
is a lead, not a proof) is reported.  With interprocedural register allocation the caller may keep values in registers the callee
is known not to touch — this finds the places where that knowledge is wrong.

    python tools/call_abi/check_call_clobbers.py listing.s [function-substring]
"""
import re
import sys
from collections import defaultdict

REG = re.compile(r'\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]')


def regs_in(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), k) for k in range(int(m.group(4)), int(m.group(5)) + 1))
    if re.search(r'\bvcc\b', text):
        out.add(('s', 'vcc'))
    return out


def split_ops(args):
    return [a.strip() for a in args.split(',')]


def parse(lines):
    """-> list of (op, writes, reads) per instruction"""
    ins = []
    for ln in lines:
        ln = ln.split(';')[0].strip()
        if not ln or ln.startswith('.') or ln.endswith(':'):
            ins.append(None)
            continue
        m = re.match(r'(\S+)\s*(.*)', ln)
        op, args = m.group(1), m.group(2)
        ops = split_ops(args) if args else []
        w, r = set(), set()
        if op.startswith(('s_waitcnt', 's_nop', 's_endpgm', 's_barrier', 's_branch', 's_cbranch', 's_setprio', 's_sleep')):
            pass
        elif op.startswith(('scratch_store', 'global_store', 'flat_store', 'ds_write', 'buffer_store', 's_cmp', 's_bitcmp', 'v_cmpx', 's_setpc')) \
                or (op.startswith('v_cmp') and op.endswith('_e32')):
            for o in ops:
                r |= regs_in(o)
            if op.startswith('v_cmp'):
                w.add(('s', 'vcc'))
        else:
            nd = 1
            if op.startswith(('v_mad_u64_u32', 'v_mad_i64_i32')) or re.match(r'v_(add|sub|subrev)_co_u32', op) or re.match(r'v_(addc|subb|subbrev)_co_u32', op) \
                    or op.startswith('v_div_scale'):
                nd = 2
            if op.startswith('s_swappc'):
                nd = 1
            for k, o in enumerate(ops):
                (w if k < nd else r).update(regs_in(o))
            if op.startswith(('v_readlane', 'v_readfirstlane')):
                pass
            if op.startswith('v_writelane'):   # partial write: also a read
                r |= regs_in(ops[0])
        ins.append((op, w, r, ln))
    return ins


def main():
    path = sys.argv[1]
    only = sys.argv[2] if len(sys.argv) > 2 else None
    text = open(path).read().split('\n')
    funcs, cur = {}, None
    for i, ln in enumerate(text):
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            cur = m.group(1)
            funcs[cur] = [i, None]
        if cur and ln.startswith('.Lfunc_end'):
            funcs[cur][1] = i
            cur = None
    parsed = {f: parse(text[a:b]) for f, (a, b) in funcs.items() if b}
    direct = {f: set().union(*[x[1] for x in p if x]) for f, p in parsed.items()}
    calls = defaultdict(list)   # f -> [(index, callee)]
    for f, p in parsed.items():
        a = funcs[f][0]
        # an SGPR pair that is only ever loaded with ONE symbol in this function names that symbol at every call through it
        # (the address is often materialised once, in a block that comes later in the listing than its first use)
        syms = defaultdict(set)
        for k, x in enumerate(p):
            m = re.search(r's_add_u32 s(\d+), s\d+, (_Z\w+)@rel32@lo', text[a + k])
            if m:
                syms[int(m.group(1))].add(m.group(2))
        target = {}
        for k, x in enumerate(p):
            raw = text[a + k]
            m = re.search(r's_add_u32 s(\d+), s\d+, (_Z\w+)@rel32@lo', raw)
            if m:
                target[int(m.group(1))] = m.group(2)
            m = re.search(r's_swappc_b64 s\[\d+:\d+\], s\[(\d+):\d+\]', raw)
            if m:
                r = int(m.group(1))
                t = next(iter(syms[r])) if len(syms[r]) == 1 else target.get(r, '?')
                calls[f].append((k, t))
    trans = {}
    def clob(f, seen=()):
        if f in trans:
            return trans[f]
        s = set(direct.get(f, ()))
        for _, c in calls.get(f, ()):
            if c in parsed and c not in seen:
                s |= clob(c, seen + (f,))
        trans[f] = s
        return s
    hits = 0
    for f, p in parsed.items():
        if only and only not in f:
            continue
        a = funcs[f][0]
        n = len(p)
        # basic blocks: leaders at labels and after branches
        label_at = {}
        for k in range(n):
            m = re.match(r'^(\.LBB\w+):', text[a + k])
            if m:
                label_at[m.group(1)] = k
        def succs(k):
            x = p[k]
            if not x:
                return [k + 1] if k + 1 < n else []
            op = x[0]
            tgt = re.search(r'(\.LBB\w+)', x[3])
            if op == 's_branch':
                return [label_at[tgt.group(1)]] if tgt and tgt.group(1) in label_at else []
            if op.startswith('s_cbranch'):
                out = [k + 1] if k + 1 < n else []
                if tgt and tgt.group(1) in label_at:
                    out.append(label_at[tgt.group(1)])
                return out
            if op.startswith(('s_setpc', 's_endpgm')):
                return []
            return [k + 1] if k + 1 < n else []
        live_in = [set() for _ in range(n + 1)]
        changed = True
        callee_at = dict(calls.get(f, ()))
        while changed:
            changed = False
            for k in range(n - 1, -1, -1):
                out = set()
                for t in succs(k):
                    out |= live_in[t]
                x = p[k]
                if x:
                    d, u = x[1], x[2]
                    if k in callee_at and callee_at[k] in parsed:   # a call defines what the callee writes... but only its RESULT is meaningful
                        d = set(d)
                    new = (out - d) | u
                else:
                    new = out
                if new != live_in[k]:
                    live_in[k] = new
                    changed = True
        for k, callee in calls.get(f, ()):
            if callee not in parsed:
                continue
            out = set()
            for t in succs(k):
                out |= live_in[t]
            bad = out & clob(callee)
            if 'fp_mul' in callee:   # Fp results come back in v0..v13
                bad = {b for b in bad if not (b[0] == 'v' and isinstance(b[1], int) and b[1] < 14)}
            bad = {b for b in bad if b not in (('s', 32), ('s', 33), ('s', 30), ('s', 31))}   # SP / FP are restored, the return address is the call's own
            if bad:
                hits += 1
                print(f"{f[:70]} +{k}: live across the call to {callee[:60]} but written by it: {sorted(bad, key=str)}")
    print("leads:", hits)


if __name__ == '__main__':
    main()
