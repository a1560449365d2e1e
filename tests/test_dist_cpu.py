"""Multi-GPU path on CPU: world_size-2 gloo run of the sharding + counter all-reduce that bench.py uses with RCCL, and the
rank-spawning of `bench.py --gpus N`.  The real-Engine variant (coast_bind_counters tensors, two ranks) needs a GPU and lives
in tests/test_gpu_parity.py::test_bench_two_ranks_*."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeEngine:  # stands in for coast_amd.Engine where there is no device: only the counters tensor matters to the collective
    def __init__(self, vals):
        self.counters = torch.tensor(vals, dtype=torch.int64)


def _engine_with(vals):
    """the engine whose totals are `vals`.  Always the stand-in here (the not-gpu suite must not behave differently on a box
    that happens to have a GPU); the real coast_amd.Engine -- its device-resident counters tensor, the one coast_bind_counters gave the C
    side -- goes through the same collective in tests/test_gpu_parity.py::test_bench_two_ranks_real_engines_counters_all_reduced."""
    return _FakeEngine(vals)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from coast_amd.dist import allreduce_counters, any_dwc_detected, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(1001, rank, world)
    eng = _engine_with([10 * (rank + 1), hi - lo, rank, 1])
    tot = allreduce_counters(eng, dist)
    q.put((rank, lo, hi, tot.cpu().tolist(), eng.counters.cpu().tolist(), any_dwc_detected(eng, dist)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_counter_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, tot0, loc0, det0), (r1, lo1, hi1, tot1, loc1, det1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 501, 501, 1001)          # contiguous, no overlap, first rank takes the odd one
    assert tot0 == tot1 == [30, 1001, 1, 2]                      # global sums on every rank
    assert loc0 == [10, 501, 0, 1] and loc1 == [20, 500, 1, 1]   # local totals untouched
    assert det0 and det1                                         # rank 1 saw a DWC mismatch -> the whole job knows


def test_shard_range_partitions():
    from coast_amd.dist import shard_range

    for n in (0, 1, 7, 8, 1 << 20, (1 << 28) + 5):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_identity():
    from coast_amd.dist import allreduce_counters

    eng = _engine_with([1, 2, 3, 4])
    assert allreduce_counters(eng, None).cpu().tolist() == [1, 2, 3, 4]
    live = allreduce_counters(eng, None, snapshot=False)  # no process group: the live totals themselves (bench.py's per-step read)
    assert live.data_ptr() == eng.counters.data_ptr()


def test_bench_gpus_n_spawns_n_ranks():
    """`python bench.py --gpus 2` must start two ranks itself (it used to run one and print n_gpus 1).  Without
    a GPU every rank stops at the same loud error -- two of them, each with its own RANK, prove the spawn."""
    import subprocess

    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["COAST_BENCH_ECHO_RANK"] = "1"
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by the gpu-marked two-rank tests")
    for attempt in range(3):  # the rendezvous of two fresh interpreters is occasionally refused on a loaded build container
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, env=env, timeout=600)
        err = p.stderr + p.stdout
        if "rank 0/2" in err and "rank 1/2" in err and err.count("no GPU visible") >= 2:
            break
    assert p.returncode != 0
    assert "rank 0/2" in err and "rank 1/2" in err, err[:3000]
    # both ranks normally get to the error; the launcher may SIGTERM the slower one the moment the first one fails
    assert err.count("no GPU visible") >= 1, err[-2000:]


def test_bench_rejects_mismatched_world():
    import subprocess

    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=env,
                       timeout=600)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def _race_builder(root, q):
    """one of the eight ranks of a multi-GPU launch on a stale tree: import the package's build module and build()"""
    sys.path.insert(0, root)
    try:
        import importlib.util

        spec = importlib.util.spec_from_file_location("race_build", os.path.join(root, "coast_amd", "build.py"))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        b._build_examples = lambda *a, **k: None  # (the C demos link against ROCm: not what this test is about)
        lib = b.build()
        q.put(("ok", b._stamp_ok(b.source_hash()), os.path.getsize(lib)))
    except Exception as e:  # noqa: BLE001
        q.put(("fail", repr(e), 0))


def test_eight_ranks_racing_build_compile_once(tmp_path):
    """the eight ranks of `bench.py --gpus 8` all come through coast_amd.build.build(); on a stale tree (sources edited,
    library from before) exactly ONE of them may compile, the others must wait on the lock and then find a current library -- never two
    compilers writing the same file, never a rank loading a half-written one.  A stand-in `hipcc` (slow, counts its invocations, writes the
    hash it was given) keeps the test at seconds; the locking and stamping logic is the real one."""
    import shutil
    import stat

    root = tmp_path / "tree"
    (root / "coast_amd" / "csrc").mkdir(parents=True)
    (root / "include").mkdir()
    shutil.copy(os.path.join(ROOT, "coast_amd", "build.py"), root / "coast_amd" / "build.py")
    (root / "include" / "coast_hip.h").write_text("/* header */\n")
    (root / "coast_amd" / "csrc" / "coast_hip.hip").write_text("// kernel source, edited\n")
    (root / "coast_amd" / "csrc" / "mm_phys_instances.hip").write_text("// second translation unit\n")
    fake = tmp_path / "bin"
    fake.mkdir()
    count = tmp_path / "compiles.txt"
    script = fake / "hipcc"
    script.write_text("""#!/bin/bash
# stand-in compiler: one line per invocation, then a slow, non-atomic write of the output (what a real link is)
echo x >> %s
out=""; hash=""
while [ $# -gt 0 ]; do
  case "$1" in -o) out="$2"; shift;; -DCOAST_SOURCE_HASH=*) hash="${1#-DCOAST_SOURCE_HASH=}";; esac
  shift
done
printf 'partial' > "$out"; sleep 1.5; printf 'library %%s' "$hash" > "$out"
""" % count)
    script.chmod(script.stat().st_mode | stat.S_IEXEC)
    lib = root / "coast_amd" / "lib"
    lib.mkdir()
    (lib / "libcoast_hip.so").write_text("library \"0000000000000000\" stale")  # the tree was edited after this was built
    old_path = os.environ["PATH"]
    os.environ["PATH"] = str(fake) + os.pathsep + old_path
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_race_builder, args=(str(root), q)) for _ in range(8)]
        for p in procs:
            p.start()
        res = [q.get(timeout=120) for _ in range(8)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        os.environ["PATH"] = old_path
    assert all(r[0] == "ok" and r[1] for r in res), res
    assert count.read_text().count("x") == 3, count.read_text()  # ONE build for eight ranks: two translation units side by side + the link
    assert len({r[2] for r in res}) == 1  # everybody saw the finished file
