"""Loaders for the committed golden fixtures (tests/golden/*.json, made by make_golden.py)."""
import gzip
import json
import os

from vechat_amd import capi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_kats():
    return json.load(open(os.path.join(GOLDEN, "spoa_kat.json")))


def load_sample_reads():
    rec = gzip.open(os.path.join(GOLDEN, "sample.fastq.gz")).read().split(b"\n")
    seqs = [rec[i + 1] for i in range(0, len(rec) - 1, 4)]
    quals = [rec[i + 3] for i in range(0, len(rec) - 1, 4)]
    return seqs, quals


def load_windows():
    return json.load(open(os.path.join(GOLDEN, "windows.json")))


def fixture_batch(wins):
    """list of fixture dicts -> one Batch (layers rank-sorted by the host helper, fasta flag from
    the host's restatement of window.cpp:223 on the fixture's quality buffer)."""
    host = capi.load_host()
    ws, fl = [], []
    for win in wins:
        bb = win["backbone"].encode()
        bq = win["backbone_quality"].encode()
        seqs = [bb] + [l["seq"].encode() for l in win["layers"]]
        quals = [bq[:len(bb)]] + [None if l["qual"] is None else l["qual"].encode() for l in win["layers"]]
        b = [0] + [l["begin"] for l in win["layers"]]
        e = [0] + [l["end"] for l in win["layers"]]
        ws.append((seqs, quals, b, e))
        fl.append(host.vc_backbone_is_fasta(bq, len(bb)))
    return capi.Batch.from_windows(ws, fl, host=host)


def load_plumbing():
    """BASELINE config A stand-in (tests/golden/make_plumbing.py): targets, reads, overlaps with CIGAR and the
    real reference's per-window results; -> (fixture dict, WindowBuilder with everything added)."""
    from vechat_amd.windows import WindowBuilder
    fx = json.load(gzip.open(os.path.join(GOLDEN, "plumbing.json.gz"), "rt"))
    wb = WindowBuilder(fx["window_length"], fx["quality_threshold"])
    for name, d, q in fx["sequences"]:
        wb.add_sequence(name, d.encode(), None if q is None else q.encode())
    wb.set_targets(fx["n_targets"])
    for o in fx["overlaps"]:
        wb.add_overlap(*o)
    return fx, wb
