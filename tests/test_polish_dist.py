"""CPU suite: the command line's one-process-per-GPU orchestration (targets split by estimated work before loading, per-rank
window building, stitched FASTA text gathered to rank 0 point to point) over gloo, world size 2.  The device is not
available here, so the consensus engine is replaced IN THIS TEST by the checker (oracle); what is verified is that two
ranks together write exactly what one rank writes, and that this equals the reference's result for the fixture."""
import io
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import fixtures
import oracle_api as oa
from test_seqio import write_inputs
from vechat_amd import capi


class OracleContext:
    """Stand-in for vechat_amd.engine.HipContext in this test only."""

    def __init__(self, device=0, **kw):
        self.params = capi.default_params(**kw)

    def consensus(self, batch, retry_overflow=True):
        cons, pol, _ = oa.oracle_run(batch, self.params)
        return cons, np.array([capi.VC_WIN_OK if p else capi.VC_WIN_UNPOLISHED for p in pol], dtype=np.uint8)

    def close(self):
        pass


def _run(rank, world, port, argv, q):
    from vechat_amd import polish
    polish.HipContext = OracleContext
    if world > 1:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                          VC_DIST_BACKEND="gloo")
    out, old = io.StringIO(), sys.stdout
    sys.stdout = out
    try:
        rc = polish.main(argv)
    finally:
        sys.stdout = old
    q.put((rank, rc, out.getvalue()))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("flags,key", [(["-p"], "hap"), ([], "linear")])
def test_two_ranks_write_what_one_rank_writes(built, tmp_path, flags, key):
    fx, wb = fixtures.load_plumbing()
    wb.close()
    rp, op, tp = write_inputs(fx, tmp_path, sam=True)
    argv = [str(rp), str(op), str(tp)] + flags
    ctx = mp.get_context("spawn")
    results = {}
    for world in (1, 2):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        q = ctx.Queue()
        procs = [ctx.Process(target=_run, args=(r, world, port, argv, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = dict((r, (rc, text)) for r, rc, text in (q.get(timeout=240) for _ in range(world)))
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        assert all(rc == 0 for rc, _ in got.values())
        if world == 2:
            assert got[1][1] == ""                               # only rank 0 writes
        results[world] = got[0][1]
    assert results[1] == results[2]
    lines = results[2].strip().split("\n")
    assert [[lines[i][1:], lines[i + 1]] for i in range(0, len(lines), 2)] == fx["expected"][key]["stitched"]
