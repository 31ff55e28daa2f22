import sys, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import torch
import __graft_entry__ as ge
from oracle import coracle as co
pkg = ge.load_package()
g = sys.argv[1]; ln = int(sys.argv[2]); n = 1 << ln
aff, jb = (96, 144) if g == "g1" else (192, 288)
bases = co.gen_bases(g, 4242, 1 << 12, 16)
one = bytes.fromhex("fdff02000000097602000cc40b00f4ebba58c7535798485f455752705358ce776dec56a2971a075c93e480fac35ef615")
one = one if g == "g1" else one + bytes(48)
m = 1 << 11
jac = b"".join(co.sum_jac(g, bases[aff * i:aff * (i + 1)] + one + bases[aff * (i + 1):aff * (i + 2)] + one, 2) for i in range(m)) * (n // m)
d_in = torch.frombuffer(bytearray(jac), dtype=torch.uint8).cuda()
d_out = torch.empty(n * aff, dtype=torch.uint8, device="cuda")
with pkg.Context([0]) as c:
    want = c.normalize_batch(g, jac[:jb * 4096])
    for _ in range(3): c.normalize_batch_device(g, d_in.data_ptr(), n, d_out.data_ptr())
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(10):
        t0 = time.perf_counter(); c.normalize_batch_device(g, d_in.data_ptr(), n, d_out.data_ptr()); best = min(best, time.perf_counter() - t0)
    p = c.profile()
    ok = bytes(d_out[:aff * 4096].cpu().numpy()) == bytes(want)
    print(json.dumps({"group": g, "log_n": ln, "device_call_ms": round(best * 1e3, 3), "kernels_ms": round(p["accumulate_ms"], 3), "ok": ok}))
