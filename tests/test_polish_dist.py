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

    def __init__(self, device=0, reserve=None, **kw):
        self.params = capi.default_params(**kw)

    @classmethod
    def in_background(cls, **kw):                       # (polish starts its context while it parses: HipContext.in_background)
        from concurrent.futures import Future
        f = Future()
        f.set_result(cls(**kw))
        return f

    def set_window_type(self, window_type):
        self.params.window_type = int(window_type)

    def reserve(self, nbytes=0):
        pass

    def consensus(self, batch, retry_overflow=True):
        cons, pol, _ = oa.oracle_run(batch, self.params)
        return cons, np.array([capi.VC_WIN_OK if p else capi.VC_WIN_UNPOLISHED for p in pol], dtype=np.uint8)

    def consensus_batched(self, batch, retry_overflow=True, fill=None, **kw):      # (the command line hands the batch over laid out, not yet written)
        if fill is not None:
            fill(0, batch.n_windows)
        return self.consensus(batch, retry_overflow=retry_overflow)

    def close(self):
        pass


def _run(rank, world, port, argv, q, py_readers=False):
    from vechat_amd import polish
    polish.HipContext = OracleContext
    if py_readers:
        os.environ["VC_PY_PARSERS"] = "1"
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
@pytest.mark.parametrize("py_readers", [False, True])
@pytest.mark.parametrize("flags,key", [(["-p", "-d", "0.2", "-s", "0.2"], "hap"), ([], "linear")])
def test_two_ranks_write_what_one_rank_writes(built, tmp_path, flags, key, py_readers):
    """py_readers False: a rank plans and loads through the C++ readers (vc_io_target_cost / vc_io_rank_names / vc_io_load); True: the
    Python readers (VC_PY_PARSERS=1), the second restatement -- the same text either way."""
    fx, wb = fixtures.load_plumbing()
    wb.close()
    rp, op, tp = write_inputs(fx, tmp_path, sam=True)
    argv = [str(rp), str(op), str(tp)] + flags
    ctx = mp.get_context("spawn")
    results = {}
    for world in (1, 2):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        q = ctx.Queue()
        procs = [ctx.Process(target=_run, args=(r, world, port, argv, q, py_readers)) for r in range(world)]
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


def _spawn(world, argv):
    ctx = mp.get_context("spawn")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    q = ctx.Queue()
    procs = [ctx.Process(target=_run, args=(r, world, port, argv, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict((r, (rc, text)) for r, rc, text in (q.get(timeout=240) for _ in range(world)))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return got


@pytest.mark.timeout(300)
def test_unpolished_targets_of_a_rank_without_overlaps(built, tmp_path):
    """-u with one rank's targets keeping no overlap at all (such a rank used to send nothing, so the N-rank output lost
    targets the single-rank run prints).  The reference builds windows for every target (polisher.cpp:389-411): a target without
    overlaps comes out unpolished under -u, from whichever rank owns it."""
    fx, wb = fixtures.load_plumbing()
    wb.close()
    rp, op, tp = write_inputs(fx, tmp_path, sam=True)
    last = fx["sequences"][fx["n_targets"] - 1][0]
    kept = [l for l in open(op) if l.startswith("@") or l.split("\t")[2] != last]      # no overlap lands on the last target
    open(op, "w").writelines(kept)
    argv = [str(rp), str(op), str(tp), "-u"]
    one, two = _spawn(1, argv), _spawn(2, argv)
    assert one[0][0] == 0 and all(rc == 0 for rc, _ in two.values())
    assert two[1][1] == "" and one[0][1] == two[0][1]
    names = [l[1:].split()[0] for l in two[0][1].split("\n") if l.startswith(">")]
    assert len(names) == fx["n_targets"] and names[-1].startswith(last)            # the overlap-less target is there, last, unpolished
    # without -u it is dropped by one rank and by two alike
    one, two = _spawn(1, argv[:-1]), _spawn(2, argv[:-1])
    assert one[0][1] == two[0][1] and len([l for l in two[0][1].split("\n") if l.startswith(">")]) == fx["n_targets"] - 1


@pytest.mark.timeout(300)
def test_a_failing_rank_ends_every_rank(built, tmp_path):
    """An exception on one rank before the gather must not leave the others waiting in it: error flags are exchanged first
    and every rank returns non-zero."""
    fx, wb = fixtures.load_plumbing()
    wb.close()
    rp, op, tp = write_inputs(fx, tmp_path, sam=True)
    argv = [str(rp), str(op), str(tp), "-p", "-w", "0"]                                 # window length 0: the builder refuses it on every rank
    got = _spawn(2, argv)
    assert all(rc != 0 for rc, _ in got.values()) and all(text == "" for _, text in got.values())
