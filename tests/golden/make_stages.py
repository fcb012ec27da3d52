"""Generates tests/golden/stages.json: per-stage digests (graph after every layer, after every prune / AddWeights round, the
final alignment) of selected windows of tests/golden/windows.json, taken from the REAL reference compiled in place
(oracle/_ref, ref_harness.cpp:vcref_window_stages -- a step-by-step replay of Window::generate_consensus with the reference's
own Graph / AlignmentEngine; the replay's consensus must equal vcref_window's, checked here).  Build container only.

  python tests/golden/make_stages.py"""
import ctypes as C
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_api as oa  # noqa: E402

NAMES = ["partial_mix/0", "two_haplotypes/1", "mixed_fasta_fastq_layers/0", "pacbio_500x32/0", "empty_final_alignment/0", "iupac_columns/0"]


def reference_stages(win, prm):
    lib = oa.load_ref("sse41")
    layers = win["layers"]
    n = len(layers)
    SA, U = C.c_char_p * max(n, 1), C.c_uint32 * max(n, 1)
    sa = SA(*[l["seq"].encode() for l in layers])
    qa = SA(*[None if l["qual"] is None else l["qual"].encode() for l in layers])
    la, ba, ea = U(*[len(l["seq"]) for l in layers]), U(*[l["begin"] for l in layers]), U(*[l["end"] for l in layers])
    cap = 8 * (3 * n + 64)
    rec = (C.c_uint64 * cap)()
    nrec, olen = C.c_uint32(0), C.c_uint32(0)
    out = C.create_string_buffer(1 << 16)
    lib.vcref_window_stages.restype = C.c_int
    rc = lib.vcref_window_stages(C.c_char_p(win["backbone"].encode()), C.c_uint32(len(win["backbone"])), C.c_char_p(win["backbone_quality"].encode()),
                                 C.c_uint32(n), sa, la, qa, ba, ea, C.c_int(prm["match"]), C.c_int(prm["mismatch"]), C.c_int(prm["gap"]),
                                 C.c_double(prm["min_confidence"]), C.c_double(prm["min_support"]), C.c_uint32(prm["num_prune"]),
                                 rec, C.c_uint32(cap), C.byref(nrec), out, C.c_uint32(1 << 16), C.byref(olen))
    assert rc == 0, rc
    return [[int(rec[8 * i + k]) for k in range(8)] for i in range(nrec.value)], out.raw[:olen.value].decode()


def main():
    gold = json.load(open(os.path.join(HERE, "windows.json")))
    by_name = {w["name"]: w for w in gold["windows"]}
    out = {}
    for name in NAMES:
        win = by_name[name]
        recs, cons = reference_stages(win, gold["params"])
        assert cons == win["expected"]["hap"]["consensus"], name      # the replay IS the reference's run
        out[name] = [[r[0], r[1], r[2], r[3], f"{r[4]:016x}", f"{r[5]:016x}", r[6], f"{r[7]:016x}"] for r in recs]
        print(name, len(recs), "stages")
    json.dump(dict(record="kind, index, nodes, edges, hash(nodes), hash(edges), pairs, hash(pairs); kinds: 1 after layer, 2 after prune, "
                          "3 after AddWeights round, 4 final alignment", windows=out), open(os.path.join(HERE, "stages.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
