"""the two-round driver (`python -m vechat_amd.driver`, reference: scripts/vechat).  CPU tests follow the
command lines and the file hand-off with stand-ins for the external tools; the GPU test runs both rounds on the device with
a stub overlapper and checks that the reads come out closer to their haplotypes."""
import json
import os
import sys

import numpy as np
import pytest

from vechat_amd import driver

HERE = os.path.dirname(os.path.abspath(__file__))
STUBS = os.path.join(HERE, "stubs")
OVL = f"{sys.executable} {os.path.join(STUBS, 'stub_overlapper.py')} {{targets}} {{reads}} {{out}}"
POL = f"{sys.executable} {os.path.join(STUBS, 'stub_polisher.py')}"


def simulate(path, n_reads=12, genome_len=3000, read_len=1800, err=0.10, seed=5, fastq=True):
    """Reads from two haplotypes (SNPs every ~150 bp) of a random genome, both strands, names <id>_<start>_<end>_<strand>."""
    rng = np.random.default_rng(seed)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    g = rng.choice(np.frombuffer(b"ACGT", np.uint8), genome_len)
    haps = [g.copy(), g.copy()]
    for p in range(75, genome_len, 150):
        haps[1][p] = rng.choice([c for c in b"ACGT" if c != g[p]])
    truth, recs = {}, []
    for k in range(n_reads):
        h = k % 2
        s = int(rng.integers(0, genome_len - read_len + 1)) if k >= 2 else 0
        e = s + read_len
        src = haps[h][s:e]
        out = bytearray()
        for c in src:
            r = rng.random()
            if r < err * 0.3:
                continue
            if r < err * 0.7:
                out.append(int(rng.choice(np.frombuffer(b"ACGT", np.uint8)))); out.append(int(c))
            elif r < err:
                out.append(int(rng.choice([x for x in b"ACGT" if x != c])))
            else:
                out.append(int(c))
        strand = "+" if k % 3 else "-"
        seq, tr = bytes(out), bytes(src)
        if strand == "-":
            seq = bytes(comp[c] for c in reversed(seq)); tr = bytes(comp[c] for c in reversed(tr))
        name = f"h{h}r{k}_{s}_{e}_{strand}"
        truth[name] = tr
        recs.append((name, seq))
    with open(path, "w") as f:
        for name, seq in recs:
            if fastq:
                f.write(f"@{name}\n{seq.decode()}\n+\n{'5' * len(seq)}\n")
            else:
                f.write(f">{name}\n{seq.decode()}\n")
    return recs, truth


def run_driver(tmp_path, monkeypatch, extra, fastq=True, n_reads=6):
    reads = tmp_path / ("reads.fastq" if fastq else "reads.fasta")
    recs, _ = simulate(str(reads), n_reads=n_reads, fastq=fastq)
    log = tmp_path / "polisher.log"
    monkeypatch.setenv("VC_STUB_LOG", str(log))
    out = tmp_path / "out.fa"
    rc = driver.main([str(reads), "-o", str(out), "--workdir", str(tmp_path / "work"), "--overlapper-r1", OVL, "--overlapper-r2", OVL,
                      "--polisher", POL] + extra)
    assert rc == 0
    calls = [json.loads(l) for l in open(log)]
    return recs, calls, out, tmp_path / "work", reads


# ---- N4 pinned against the reference: tests/golden/driver_cmds.json was recorded from scripts/vechat ITSELF (make_driver.py runs it
# in place with its shell-outs captured); every scenario below replays the same argument vector through vechat_amd.driver.
FIX = json.load(open(os.path.join(HERE, "golden", "driver_cmds.json")))
TOOL_HEADS = ("minimap2", "yacrd", "{RACON}")


def _norm(cmd):
    """One command line, comparable across the two drivers: directories dropped from file arguments, whitespace collapsed."""
    out = []
    for tok in cmd.split():
        lead = ">" if tok.startswith(">") and len(tok) > 1 else ""
        body = tok[len(lead):].strip("'\"")
        if "|" not in tok and (body.startswith("/") or body.startswith("{CWD}/")) and not body.startswith("/dev/"):
            tok = lead + os.path.basename(body)
        out.append(tok)
    return " ".join(out).replace("| ", "|").replace(" |", "|")       # `a| b` and `a|b` are the same pipeline


@pytest.fixture
def tool_stand_ins(tmp_path, monkeypatch):
    """minimap2 / fpa / yacrd stand-ins first on PATH -- the same ones make_driver.py gave the reference script: minimap2 prints an
    all-vs-all PAF of the two files it is given (wide enough to pass the awk filters), fpa passes its input through, yacrd copies
    the reads to the scrubbed file."""
    bindir = tmp_path / "bin"
    bindir.mkdir()
    (bindir / "minimap2").write_text(f"""#!{sys.executable}
import gzip, os, sys
files = [a for a in sys.argv[1:] if os.path.isfile(a)]
def records(path):
    lines = [l.rstrip("\\n") for l in (gzip.open(path, "rt") if path.endswith(".gz") else open(path))]
    per = 4 if lines and lines[0].startswith("@") else 2
    return [(lines[i][1:].split()[0], lines[i + 1]) for i in range(0, len(lines) - 1, per)]
for qn, qs in records(files[1]):
    for tn, ts in records(files[0]):
        if qn != tn:
            print("\\t".join(map(str, [qn, len(qs), 0, len(qs), "+", tn, len(ts), 0, len(ts), 5000, 5000, 60])))
""")
    (bindir / "fpa").write_text("#!/bin/bash\ncat\n")
    (bindir / "yacrd").write_text("""#!/bin/bash
args=("$@"); for ((i=0;i<${#args[@]};i++)); do if [ "${args[$i]}" = "scrubb" ]; then in="${args[$((i+2))]}"; out="${args[$((i+4))]}"; fi; done
cp "$in" "$out"
""")
    for f in ("minimap2", "fpa", "yacrd"):
        os.chmod(bindir / f, 0o755)
    monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("VC_STUB_LOG", str(tmp_path / "polisher.log"))
    monkeypatch.setenv("VC_STUB_TAG", "r")
    return bindir


@pytest.mark.parametrize("name", sorted(FIX["scenarios"]))
def test_command_lines_and_hand_off_match_the_reference_script(name, tmp_path, monkeypatch, tool_stand_ins):
    sc = FIX["scenarios"][name]
    reads = tmp_path / ("reads.fastq" if sc["fastq"] else "reads.fasta")
    with open(reads, "w") as f:
        for n, s in sc["reads"]:
            f.write(f"@{n}\n{s}\n+\n{'5' * len(s)}\n" if sc["fastq"] else f">{n}\n{s}\n")
    work = tmp_path / "work"
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(driver, "COMMAND_LOG", [])
    assert driver.main([str(reads)] + sc["argv"] + ["--workdir", str(work), "--polisher", POL]) == 0
    ours = [_norm(c.replace(POL, "{RACON}")) for c in driver.COMMAND_LOG]
    theirs = [_norm(c) for c in sc["commands"] if c.split()[0] in TOOL_HEADS]
    assert ours == theirs
    # the polisher really received those arguments (the log of the stand-in), in the reference's order
    calls = [json.loads(l) for l in open(tmp_path / "polisher.log")]
    assert [" ".join(["{RACON}"] + [os.path.basename(x) if x.startswith("/") else x for x in c]) for c in calls] == \
           [" ".join(t.split(">")[0].split()) for t in theirs if t.startswith("{RACON}")]
    outname = sc["argv"][sc["argv"].index("-o") + 1] if "-o" in sc["argv"] else "reads.corrected.fa"
    assert open(tmp_path / outname).read() == sc["output"]
    # what stays behind: the reference works in the current directory, this driver in --workdir
    left = sorted(set(os.listdir(work)) | {outname, reads.name})
    assert left == sorted(f for f in sc["left_behind"])


def test_failing_overlapper_stops_the_round(tmp_path, monkeypatch, tool_stand_ins):
    """a missing / failing minimap2 must not surface rounds later as an empty overlap set (the pipelines run under pipefail)."""
    (tool_stand_ins / "minimap2").write_text("#!/bin/bash\nexit 7\n")
    reads = tmp_path / "reads.fastq"
    simulate(str(reads), n_reads=4)
    with pytest.raises(RuntimeError, match="command failed"):
        driver.main([str(reads), "-o", str(tmp_path / "o.fa"), "--workdir", str(tmp_path / "work"), "--polisher", POL])


def test_split_reads_gzip_input(tmp_path, monkeypatch, tool_stand_ins):
    """--split narrows the query file from a gzipped input as well (the reference's plain open() would fail there)."""
    import gzip
    sc = FIX["scenarios"]["split_fastq"]
    reads = tmp_path / "reads.fastq.gz"
    with gzip.open(reads, "wt") as f:
        for n, s in sc["reads"]:
            f.write(f"@{n}\n{s}\n+\n{'5' * len(s)}\n")
    monkeypatch.chdir(tmp_path)
    assert driver.main([str(reads)] + sc["argv"] + ["--workdir", str(tmp_path / "work"), "--polisher", POL]) == 0
    assert open(tmp_path / "reads.corrected.fa").read() == sc["output"]


def test_pass_through_values_must_be_numbers_and_keep_going_reaches_the_polisher(tmp_path, monkeypatch, tool_stand_ins):
    """the pass-through options are formatted into `bash -c` command lines, so only plain numbers get that far; and a
    single uncomputable window must not have to end a two-round run: --keep-going (or VC_KEEP_GOING=1) is handed to our polisher."""
    reads = tmp_path / "reads.fastq"
    simulate(str(reads), n_reads=4)
    for bad in (["-t", "1; touch pwned"], ["-d", "0.2$(id)"], ["--min-ovlplen-cns", "1e3 x"], ["--platform", "pb; id"]):
        with pytest.raises(SystemExit):
            driver.main([str(reads), "--workdir", str(tmp_path / "w")] + bad)
    assert not (tmp_path / "pwned").exists()
    # our own polisher command (stubbed by a recorder in front of it): the flag is appended in both rounds
    seen = []
    monkeypatch.setattr(driver, "_sh", lambda cmd, cwd=None: (seen.append(cmd), (_ for _ in ()).throw(RuntimeError("stop")))[0]
                        if "vechat_amd.polish" in cmd else None)
    monkeypatch.chdir(tmp_path)
    with pytest.raises(RuntimeError, match="stop"):
        driver.main([str(reads), "--workdir", str(tmp_path / "w2"), "--keep-going", "-d", "0.25"])
    assert seen and seen[0].split(" -m vechat_amd.polish ")[1].startswith("-f -p -d 0.25 -s 0.2 -t 1 --keep-going ")
    monkeypatch.setenv("VC_KEEP_GOING", "1")
    seen.clear()
    with pytest.raises(RuntimeError, match="stop"):
        driver.main([str(reads), "--workdir", str(tmp_path / "w3"), "--linear"])
    assert "--keep-going" in seen[0]


def test_helpers(tmp_path):
    p = tmp_path / "x.fa"
    p.write_text(">a\nAC\n>b\nGT\n>c\nAA\n")
    assert driver.fq_or_fa(str(p)) == "fa"
    chunks = driver.split_lines(str(p), 4, "fa", str(tmp_path))
    assert [os.path.basename(c) for c in chunks] == ["reads_chunk00.fa", "reads_chunk01.fa"]
    assert open(chunks[1]).read() == ">c\nAA\n"
    q = tmp_path / "x.fq"
    q.write_text("@a\nAC\n+\n!!\n")
    assert driver.fq_or_fa(str(q)) == "fq"
    with pytest.raises(ValueError):
        (tmp_path / "bad").write_text("hello\n")
        driver.fq_or_fa(str(tmp_path / "bad"))


def _fit_distance(a, b):
    """Edit distance of `a` against the best-matching substring of `b` (a corrected read may have lost its ends to windows
    without coverage: what counts is the error rate of what is there)."""
    a, b = np.frombuffer(a, np.uint8), np.frombuffer(b, np.uint8)
    prev = np.zeros(len(b) + 1, np.int64)
    for i in range(1, len(a) + 1):
        cur = np.empty_like(prev)
        cur[0] = i
        cur[1:] = np.minimum(prev[:-1] + (b != a[i - 1]), prev[1:] + 1)
        for j in range(1, len(b) + 1):                        # horizontal pass
            if cur[j - 1] + 1 < cur[j]:
                cur[j] = cur[j - 1] + 1
        prev = cur
    return int(prev.min())


@pytest.mark.gpu
def test_both_rounds_on_the_device(built, tmp_path):
    """reads.fastq -> round 1 (haplotype-aware, -f -p -d 0.2 -s 0.2) -> round 2 (linear, -f) through the real polisher; overlaps
    from the stub overlapper (plain PAF, aligned on the device).  Every read is corrected and ends up much closer to the
    haplotype it was drawn from (error rate of the corrected bases, ends lost to uncovered windows not counted)."""
    reads = tmp_path / "reads.fastq"
    recs, truth = simulate(str(reads), n_reads=16, genome_len=2400, read_len=1500, err=0.10, seed=9)
    out = tmp_path / "out.fa"
    rc = driver.main([str(reads), "-o", str(out), "--workdir", str(tmp_path / "work"), "--overlapper-r1", OVL, "--overlapper-r2", OVL + " 400",
                      "-u"])
    assert rc == 0
    lines = open(out).read().split("\n")
    got = {lines[i][1:].split()[0].rstrip("r"): lines[i + 1].encode() for i in range(0, len(lines) - 1, 2)}    # one 'r' per round
    assert set(got) == {n for n, _ in recs}
    picks = recs[:6]
    before = sum(_fit_distance(s, truth[n]) for n, s in picks) / sum(len(s) for _, s in picks)
    after = sum(_fit_distance(got[n], truth[n]) for n, _ in picks) / sum(len(got[n]) for n, _ in picks)
    assert all(len(got[n]) > 0.7 * len(truth[n]) for n, _ in picks)
    assert after * 3 < before, (before, after)               # ~10 % raw error rate


@pytest.mark.gpu
def test_two_rounds_on_the_device_match_the_reference(built, tmp_path):
    """Row N4 as a parity test: tests/golden/two_rounds.json.gz (make_two_rounds.py) holds a read set, the overlaps an external
    overlapper would deliver for round 1 and -- computed on the reference's round-1 result -- for round 2, and both rounds' FASTA
    with every window's consensus taken from the reference itself (oracle/_ref: window.cpp, both overloads).  The driver runs the
    two rounds through the real polisher on the device; the final text must be identical (it can only be if round 1 was, too:
    round 2's overlaps address the reference's round-1 sequences base by base)."""
    import gzip
    fx = json.load(gzip.open(os.path.join(os.path.dirname(__file__), "golden", "two_rounds.json.gz"), "rt"))
    reads = tmp_path / "reads.fastq"
    reads.write_text(fx["reads_fastq"])
    for k in ("round1_paf", "round2_paf"):
        (tmp_path / (k + ".paf")).write_text(fx[k])
    out = tmp_path / "out.fa"
    rc = driver.main([str(reads), "-o", str(out), "--workdir", str(tmp_path / "work"),
                      "--overlapper-r1", f"cp {tmp_path / 'round1_paf.paf'} {{out}}", "--overlapper-r2", f"cp {tmp_path / 'round2_paf.paf'} {{out}}"])
    assert rc == 0
    assert open(out).read() == fx["round2_fasta"]
    # the same two rounds inside ONE process on ONE context (--in-process: vc_set_polish_params switches the overload between the rounds)
    out2 = tmp_path / "out_in_process.fa"
    rc = driver.main([str(reads), "-o", str(out2), "--workdir", str(tmp_path / "work2"), "--in-process",
                      "--overlapper-r1", f"cp {tmp_path / 'round1_paf.paf'} {{out}}", "--overlapper-r2", f"cp {tmp_path / 'round2_paf.paf'} {{out}}"])
    assert rc == 0
    assert open(out2).read() == fx["round2_fasta"]
    assert not driver._SHARED                        # the shared context went with the run
    # and the haplotype-aware round alone (--linear would be the other overload; here: one explicit polisher call on round 1)
    from vechat_amd import polish
    import io
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert polish.main(["-f", "-p", "-d", "0.2", "-s", "0.2", str(reads), str(tmp_path / "round1_paf.paf"), str(reads)]) == 0
    assert buf.getvalue() == fx["round1_fasta"]
