"""The measured-count files the bench prices its roofline with belong to the kernel sources in the tree (bench.kernel_hash): a kernel change without
a new PMC round would otherwise quote the old instruction counts.  And build()'s warning gate counts what it should."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_profile_counts_belong_to_these_kernel_sources():
    import bench
    h = bench.kernel_hash()
    traffic = json.load(open(os.path.join(ROOT, "profiles", "r6_hbm_traffic.json")))
    mix = json.load(open(os.path.join(ROOT, "profiles", "r6_valu_mix.json")))
    assert traffic["kernel_hash"] == h, "profiles/r6_hbm_traffic.json was measured on other kernel sources: run tools/gpu_round.sh on the GPU box and copy its output"
    assert mix["kernel_hash"] == h, "profiles/r6_valu_mix.json is of other kernel sources: python tools/valu_mix.py"
    line = json.loads(open(os.path.join(ROOT, "profiles", "r6_bench.json")).read().strip().splitlines()[-1])
    assert line["roofline"]["kernel_hash"] == h
    # the bench line carries what the contract asks for
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["roofline"]["bound"] in ("hbm", "mfma", "valu_issue") and 0.0 < line["roofline"]["frac"] <= 1.0
    assert line["cpu_baseline"]["kind"] == "reference" and line["cpu_baseline"]["parity_mismatches"] == 0


def test_build_counts_compiler_warnings(capfd):
    import __graft_entry__ as g
    before = len(g._WARNINGS)
    g._run([sys.executable, "-c", "import sys; sys.stderr.write('x.hip:1:1: warning: something [-Wsomething]\\n')"])
    assert len(g._WARNINGS) == before + 1
    g._run([sys.executable, "-c", "import sys; sys.stderr.write('ref.cpp:1:1: warning: theirs\\n')"], count=False)      # (the reference's sources under oracle/_ref)
    assert len(g._WARNINGS) == before + 1
    del g._WARNINGS[before:]
    try:
        g._run([sys.executable, "-c", "raise SystemExit(3)"])
    except subprocess.CalledProcessError as e:
        assert e.returncode == 3
    else:
        raise AssertionError("a failing build step must raise")
