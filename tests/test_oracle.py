"""CPU suite, part 1: the oracle (oracle/vc_oracle.c) against the golden vectors.

  * the four linear-gap known-answer tests of the reference's own suite
    (vendor/spoa/test/spoa_test.cpp: Local, LocalWithQualities, Global, GlobalWithQualities),
  * window fixtures whose expected outputs were produced by the REAL reference (tests/golden),
  * and, where oracle/_ref is present, live differential runs against the reference itself.
"""
import pytest

import fixtures
import oracle_api as oa
from vechat_amd import capi


@pytest.mark.parametrize("name", ["Local", "LocalWithQualities", "Global", "GlobalWithQualities"])
def test_spoa_known_answers(built, name):
    kat = fixtures.load_kats()[name]
    seqs, quals = fixtures.load_sample_reads()
    assert len(seqs) == 55                                   # spoa_test.cpp:30
    atype = {"SW": 0, "NW": 1}[kat["type"]]
    got = oa.spoa_consensus(oa.load_oracle(), "vco", seqs, quals if kat["quality"] else None,
                            atype, kat["m"], kat["n"], kat["g"])
    assert got.decode() == kat["consensus"]


@pytest.mark.parametrize("mode,key", [(0, "hap"), (1, "linear")])
def test_oracle_matches_golden_windows(built, mode, key):
    gold = fixtures.load_windows()
    batch = fixtures.fixture_batch(gold["windows"])
    p = capi.default_params(mode=mode)
    cons, pol, _ = oa.oracle_run(batch, p)
    for w, win in enumerate(gold["windows"]):
        exp = win["expected"][key]
        assert cons[w].decode() == exp["consensus"], win["name"]
        assert bool(pol[w]) == exp["polished"], win["name"]


def test_fasta_short_window_quirk_is_reproduced(built):
    """window.cpp:223 compares a C string: a short last window of a FASTA target is treated as
    FASTQ and collapses to a 1-base consensus.  Must be reproduced, not fixed."""
    gold = fixtures.load_windows()
    wins = [w for w in gold["windows"] if w["name"].startswith("fasta_short_last_window_quirk")]
    assert wins and all(len(w["expected"]["hap"]["consensus"]) == 1 for w in wins)
    batch = fixtures.fixture_batch(wins)
    assert not batch.win_fasta.any()
    cons, _, _ = oa.oracle_run(batch, capi.default_params())
    assert [len(c) for c in cons] == [1] * len(wins)


@pytest.mark.skipif(not oa.have_ref(), reason="oracle/_ref not built (reference tree absent)")
@pytest.mark.parametrize("seed,L,D,kw", [
    (31, 160, 14, dict(frac_partial=0.3)),
    (32, 220, 24, dict(n_haplotypes=2, snp_rate=0.02)),
    (33, 180, 10, dict(fastq=0, backbone_fastq=0, frac_partial=0.2)),
    (34, 500, 20, dict(profile=capi.ONT)),
])
def test_oracle_vs_reference_live(built, seed, L, D, kw):
    batch = capi.synth_batch(capi.synth_cfg(seed, L, D, **kw), 0, 3, n_threads=1)
    for mode in (0, 1):
        p = capi.default_params(mode=mode)
        cons, pol, _ = oa.oracle_run(batch, p)
        for w in range(batch.n_windows):
            for kind in ("sse41", "sisd"):
                r, rp = oa.ref_window(batch, w, p, kind=kind)
                assert r == cons[w] and rp == int(pol[w]), (mode, w, kind)


@pytest.mark.skipif(not oa.have_ref(), reason="oracle/_ref not built")
def test_alignment_and_rank_internals_match_reference(built):
    """Not just the consensus: one Align() pair list and rank_to_node of the graph."""
    seqs, quals = fixtures.load_sample_reads()
    for build_type, qtype in ((1, 1), (1, 0), (0, 0)):
        a = oa.spoa_align_probe(oa.load_oracle(), "vco", seqs[:20], quals[:20], build_type, 3, -5, -4, seqs[21], qtype)
        b = oa.spoa_align_probe(oa.load_ref(), "vcref", seqs[:20], quals[:20], build_type, 3, -5, -4, seqs[21], qtype)
        assert a == b


def test_stage_digests_match_the_reference(built):
    """SURVEY 8(c) golden intermediates: the graph after every layer, after every prune and AddWeights round, and the final
    alignment, as digests taken from the real reference (tests/golden/stages.json, generator make_stages.py).  A mismatch
    names the first stage that differs -- incl. the window whose final alignment is empty and the IUPAC one."""
    import json
    import os
    st = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stages.json")))
    gold = fixtures.load_windows()
    by_name = {w["name"]: w for w in gold["windows"]}
    assert len(st["windows"]) >= 5
    for name, exp in st["windows"].items():
        batch = fixtures.fixture_batch([by_name[name]])
        got = oa.oracle_stages(batch, capi.default_params(mode=0), 0)
        got = [[r[0], r[1], r[2], r[3], f"{r[4]:016x}", f"{r[5]:016x}", r[6], f"{r[7]:016x}"] for r in got]
        assert len(got) == len(exp), (name, len(got), len(exp))
        for k, (g, e) in enumerate(zip(got, exp)):
            assert g == e, (name, "first differing stage", k, g, e)
