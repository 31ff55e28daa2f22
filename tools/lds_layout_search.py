# brute-force check of LDS bank conflicts for ds_read_b128 in k_miller_accumulate layouts
import itertools
GROUPS_B128 = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
               list(range(4,12))+list(range(16,20))+list(range(28,32)),
               list(range(32,36))+list(range(44,48))+list(range(52,60)),
               list(range(36,44))+list(range(48,52))+list(range(60,64))]
def conflicts(addr_fn):
    """addr_fn(lane) -> word address of a 16-byte read (multiple of 4); returns extra cycles summed over the 4 lane groups"""
    extra = 0
    for grp in GROUPS_B128:
        byslot = {}
        for l in grp:
            a = addr_fn(l)
            byslot.setdefault((a // 4) % 16, set()).add(a)
        extra += max(len(v) for v in byslot.values()) - 1
    return extra
def layout_cost(R, S, PW, verbose=False):
    """two planes: U at 0, W at PW; record stride R words, group stride S words; values at 16-word steps inside a record"""
    total = 0
    for d in (2, 3):                 # terms 2 and 3 (term 1 comes from registers)
        for val in range(3):
            for q in range(4):
                def addr(l):
                    g, k = divmod(l, 6)
                    j = k - d; wrapped = j < 0
                    if wrapped: j += 6
                    return (PW if wrapped else 0) + g * S + j * R + val * 16 + 4 * q
                total += conflicts(addr)
    # squaring reads: lane reads records i and j (c0, c1 of plane U) for its terms
    sq = 0
    for t in range(4):
        for which in (0, 1):
            for val in range(2):
                for q in range(4):
                    def addr(l):
                        g, k = divmod(l, 6)
                        terms = [i for i in range(6) if i <= (k - i) % 6]
                        i = terms[t] if t < len(terms) else 0
                        j = (k - i) % 6
                        r = i if which == 0 else j
                        return g * S + r * R + val * 16 + 4 * q
                    sq += conflicts(addr)
    return total, sq
best = []
for R in range(48, 100, 4):
    for S in range(6 * R, 6 * R + 68, 4):
        for PWm in range(0, 64, 4):
            PW = 11 * S + PWm
            t, sq = layout_cost(R, S, PW)
            best.append((t, sq, R, S, PWm))
best.sort()
print(best[:15])
# current layout for reference: single plane, R = 84, variants at 0,16,32,48,64
def current():
    total = 0
    for d in (0, 2, 3):
        for val in range(3):
            for q in range(4):
                def addr(l):
                    g, k = divmod(l, 6)
                    j = k - d; wrapped = j < 0
                    if wrapped: j += 6
                    off = [(0, 48), (16, 32), (32, 64)][val][1 if wrapped else 0]
                    return g * 504 + j * 84 + off + 4 * q
                total += conflicts(addr)
    return total
print("current extra cycles per step (of", 36 * 4, "base):", current())
