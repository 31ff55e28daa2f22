"""The N>1 path on CPU: world_size-2 `gloo` run of the exchange steps (all-gather of per-rank partial sums + deterministic fold in
rank order via mi_g1_sum; and, what bench.py uses since round 3, all_gather_into_tensor of per-WINDOW sums + mi_g1_fold_windows).  The per-rank partials come from the
oracle here (no GPU); on the GPU box the same code path gathers the partials the HIP pipeline produced."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import __graft_entry__ as ge
    from oracle import coracle as co
    pkg = ge.load_package()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 300
    seed_b, seed_s = 11 + 1000 * rank, 22 + 1000 * rank
    bases = co.gen_bases("g1", seed_b, n, 1)
    sc = co.gen_scalars(seed_s, n)
    part = co.msm("g1", bases, sc, n, 0, 1)                      # stands in for the per-GPU HIP partial sum
    mine = torch.frombuffer(bytearray(part), dtype=torch.uint8)
    gather = [torch.empty(144, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(gather, mine)
    total = pkg.g1_sum([t.numpy().tobytes() for t in gather])    # identical on every rank
    # expected: closed form of every rank's shard, folded by the oracle
    exp = torch.frombuffer(bytearray(co.dlog_expected("g1", sc, seed_b, n)), dtype=torch.uint8)
    eg = [torch.empty(96, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(eg, exp)
    one = bytes.fromhex("fdff02000000097602000cc40b00f4ebba58c7535798485f455752705358ce776dec56a2971a075c93e480fac35ef615")
    want = co.sum_jac("g1", b"".join(t.numpy().tobytes() + one for t in eg), world)
    assert co.to_affine("g1", total) == co.to_affine("g1", want), "fold mismatch"
    chk = [None] * world
    dist.all_gather_object(chk, total)
    assert all(c == chk[0] for c in chk), "ranks disagree"
    # round 3: the exchange bench.py uses at N > 1 — every rank contributes its PER-WINDOW sums (here: known multiples of the
    # generator standing in for what mi_msm_g1_device_windows leaves in device memory), all_gather_into_tensor, mi_g1_fold_windows
    # on every rank (ranks added per window in rank order, then the Horner fold); expected value from the Python big-int oracle
    from oracle import bls12_381 as o
    import random
    c, nwin = 16, 16
    ks = [[random.Random(1000 * r + w).randrange(1, 1 << 60) for w in range(nwin)] for r in range(world)]
    mine_w = b"".join(o.jac_to_bytes(o.F1, o.jac_from_aff(o.F1, o.scalar_mul(o.F1, o.G1_GEN, k))) for k in ks[rank])
    gw = torch.empty(world * nwin * 144, dtype=torch.uint8)
    dist.all_gather_into_tensor(gw, torch.frombuffer(bytearray(mine_w), dtype=torch.uint8))
    folded = pkg.fold_windows("g1", gw.numpy(), world, nwin, c, nwin)
    total_k = sum((1 << (c * w)) * sum(ks[r][w] for r in range(world)) for w in range(nwin)) %% o.R_ORDER
    assert co.to_affine("g1", folded) == o.affine_to_bytes(o.F1, o.scalar_mul(o.F1, o.G1_GEN, total_k)), "window fold mismatch"
    dist.barrier()
    if rank == 0:
        print("GLOO_FOLD_OK")
""") % ROOT


def test_two_rank_gloo_allgather_and_fold(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(29300 + os.getpid() % 200), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "GLOO_FOLD_OK" in out.stdout


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no launcher in the command line and no WORLD_SIZE in the environment: bench.py spawns the
    ranks itself (a child `python -m torch.distributed.run`, started before the parent imports torch or touches a GPU) and relays
    output and exit code.  There is no GPU here, so the ranks run bench.py's launch probe (BENCH_LAUNCH_PROBE=1: meet over gloo,
    rank 0 prints one line); the same command without the probe is a `-m gpu` test (tests/test_gpu_msm.py)."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BENCH_LAUNCH_PROBE="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "3", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d == {"launch_probe": True, "n_gpus": 2, "ranks": [0, 1], "argv": ["--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "3", "--warmup", "1"]}
    # a rank that fails makes the launcher fail: without a GPU and without the probe every rank exits with bench.py's message
    env.pop("BENCH_LAUNCH_PROBE")
    import torch
    if not torch.cuda.is_available():
        bad = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert bad.returncode != 0 and "needs an MI355X" in bad.stderr
