"""CPU suite, part 3: the N>1 path (window sharding + final gather) over gloo, world_size 2."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from vechat_amd.shard import estimated_cells, gather_consensus, shard_range, shard_range_balanced


def test_shard_range_partitions_in_order():
    for n in (0, 1, 7, 8, 100003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_balanced_split_is_a_partition_with_even_cost(built):
    from vechat_amd import capi
    rng = np.random.default_rng(5)
    for world in (1, 2, 3, 8):
        cost = rng.integers(1, 100, size=257).astype(float)
        spans = [shard_range_balanced(cost, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == cost.size
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        per = [cost[a:b].sum() for a, b in spans]
        assert max(per) - min(per) <= 2 * cost.max()
    assert shard_range_balanced(np.zeros(5), 1, 2) == shard_range(5, 1, 2)
    # deep windows weigh more than shallow ones
    parts = [capi.synth_batch(capi.synth_cfg(3, 120, d), 0, 2) for d in (4, 16)]
    wins = [p.window(w) for p in parts for w in range(2)]
    b = capi.Batch.from_windows(wins, [0] * 4, presorted=True)
    c = estimated_cells(b)
    assert c[2] > 5 * c[0] and c[3] > 5 * c[1]


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(1234)
    lens_all = rng.integers(0, 40, size=n)
    blob = rng.integers(65, 85, size=int(lens_all.sum()), dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum(lens_all)])
    lo, hi = shard_range(n, rank, world)
    cons = torch.from_numpy(blob[off[lo]:off[hi]].copy())
    lens = torch.from_numpy(lens_all[lo:hi].astype(np.int64))
    c, l = gather_consensus(cons, lens, dst=0)
    if rank == 0:
        q.put((c.numpy().tobytes() == blob.tobytes(), l.numpy().tolist() == lens_all.tolist()))
    else:
        assert c is None and l is None
    # dst=None: every rank ends up with everything (used to share device-aligned CIGAR strings)
    c2, l2 = gather_consensus(cons, lens, dst=None)
    assert c2.numpy().tobytes() == blob.tobytes() and l2.numpy().tolist() == lens_all.tolist()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_over_gloo_world_size_2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 101, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=90)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert ok == (True, True)


def test_bench_gpus_flag_starts_that_many_ranks(built):
    """`python bench.py --gpus 2` itself starts two ranks (one per GPU on a GPU box).  Here: gloo, kernels stubbed off
    (VC_BENCH_STUB=1) -- the fan-out, the rendezvous, the gather to rank 0 and the one JSON line are what is checked."""
    import json, subprocess
    env = dict(os.environ, VC_BENCH_STUB="1", MASTER_PORT="29571")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--windows", "6",
                          "--length", "80"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["windows_gathered"] == 12
    # a mismatch between --gpus and the launcher's world size is an error, not a silent 1-GPU run
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)
