#!/usr/bin/env python3
"""Timeline of the LAST MSM call in a rocprofv3 --kernel-trace CSV: per dispatch the queue, start and end relative to the call's first
kernel, plus the union of the accumulate launches (the interval mi_profile.accumulate_ms reports for a pipelined call).
    python tools/trace_overlap.py <kernel_trace.csv> [window_ms]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 8e6
def short(n):
    m = re.search(r"(k_[a-z0-9_]+|__amd_rocclr_[a-zA-Z]+)", n)
    return m.group(1) if m else n[:30]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows]
ev.sort()
last_end = max(e[1] for e in ev if e[2].startswith("k_combine"))
call = [e for e in ev if e[0] >= last_end - win and e[1] <= last_end]
# the call starts at the first sort kernel after a gap of > 0.2 ms without kernels
starts = [i for i in range(1, len(call)) if call[i][0] - max(c[1] for c in call[:i]) > 200000]
if starts: call = call[starts[-1]:]
t0 = call[0][0]
for s, e, n, q in call:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q{q}  {n}")
acc = [(s, e) for s, e, n, q in call if n.startswith("k_accumulate")]
if acc:
    print(f"accumulate launches: {len(acc)}, sum {sum(e - s for s, e in acc) / 1e6:.3f} ms, union span {(max(e for s, e in acc) - min(s for s, e in acc)) / 1e6:.3f} ms")
print(f"call: {(call[-1][1] - t0) / 1e6:.3f} ms on the GPU")
