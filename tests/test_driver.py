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


def test_two_rounds_command_lines_and_hand_off(tmp_path, monkeypatch):
    recs, calls, out, work, reads = run_driver(tmp_path, monkeypatch, ["-t", "3"])
    tmp1 = str(work / "reads.corrected.tmp1.fa")
    paf = str(work / "overlap.paf")
    # round 1: vechat_racon -f -p -d 0.2 -s 0.2 -t T reads overlap.paf reads (scripts/vechat:70-72); round 2: -f -t T on round 1's output (:91-93)
    assert calls == [["-f", "-p", "-d", "0.2", "-s", "0.2", "-t", "3", str(reads), paf, str(reads)],
                     ["-f", "-t", "3", tmp1, paf, tmp1]]
    got = open(out).read().split("\n")
    assert [l[1:] for l in got[0::2] if l] == [n for n, _ in recs]
    left = sorted(os.listdir(work))
    assert not [f for f in left if f.startswith("reads.corrected.tmp") or f.startswith("reads_chunk") or f.startswith("query_sequences")], left
    assert os.path.getsize(paf) > 0                         # the stub overlapper found the simulated overlaps


def test_linear_is_one_round(tmp_path, monkeypatch):
    _, calls, _, _, reads = run_driver(tmp_path, monkeypatch, ["--linear", "-u"])
    assert len(calls) == 1 and calls[0][:4] == ["-f", "-u", "-t", "1"] and calls[0][-1] == str(reads)


@pytest.mark.parametrize("fastq", [True, False])
def test_split_chunks_targets_and_narrows_queries(tmp_path, monkeypatch, fastq):
    per = 4 if fastq else 2
    recs, calls, out, work, reads = run_driver(tmp_path, monkeypatch, ["--split", "--split-size", str(2 * per)], fastq=fastq, n_reads=6)
    # round 1: 6 records, 2 per chunk; round 2 splits the FASTA of round 1 with split_size/2 lines for FASTQ input (scripts/vechat:318-319)
    assert len(calls) == 6
    r1, r2 = calls[:3], calls[3:]
    assert all(c[:2] == ["-f", "-p"] for c in r1) and all(c[0] == "-f" and "-p" not in c for c in r2)
    assert [os.path.basename(c[-1]) for c in r1] == [f"reads_chunk{k:02d}.{'fq' if fastq else 'fa'}" for k in range(3)]
    assert [os.path.basename(c[-1]) for c in r2] == [f"reads_chunk{k:02d}.fa" for k in range(3)]
    assert all(os.path.basename(c[-3]).startswith("query_sequences.tmp.") for c in calls)      # scripts/vechat:54-57
    names = [l[1:] for l in open(out).read().split("\n")[0::2] if l]
    assert names == [n for n, _ in recs]                     # chunk outputs concatenated in order
    assert not [f for f in os.listdir(work) if f.startswith("reads_chunk") or f.startswith("reads.corrected.tmp")]


def test_scrub_and_default_overlapper_command_lines(tmp_path, monkeypatch):
    """--scrub (scripts/vechat:189-205) and the default minimap2 | awk | fpa pipelines (:36-49), followed through stand-ins for
    the external binaries put first on PATH: each records its arguments, minimap2 prints the stub overlapper's PAF, fpa passes
    its input through, yacrd copies the reads to the scrubbed file."""
    bindir = tmp_path / "bin"
    bindir.mkdir()
    log = tmp_path / "tools.log"
    stub = os.path.join(STUBS, "stub_overlapper.py")
    (bindir / "minimap2").write_text(f"""#!/bin/bash
echo "minimap2 $@" >> {log}
args=("$@"); n=${{#args[@]}}
t=""; q=""
for a in "${{args[@]}}"; do if [ -f "$a" ]; then if [ -z "$t" ]; then t="$a"; else q="$a"; fi; fi; done
{sys.executable} {stub} "$t" "$q" /dev/stdout 300
""")
    (bindir / "fpa").write_text(f"#!/bin/bash\necho \"fpa $@\" >> {log}\ncat\n")
    (bindir / "yacrd").write_text(f"""#!/bin/bash
echo "yacrd $@" >> {log}
args=("$@"); for ((i=0;i<${{#args[@]}};i++)); do if [ "${{args[$i]}}" = "scrubb" ]; then in="${{args[$((i+2))]}}"; out="${{args[$((i+4))]}}"; fi; done
cp "$in" "$out"
""")
    for f in ("minimap2", "fpa", "yacrd"):
        os.chmod(bindir / f, 0o755)
    monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("VC_STUB_LOG", str(tmp_path / "polisher.log"))
    reads = tmp_path / "reads.fastq"
    recs, _ = simulate(str(reads), n_reads=6)
    out = tmp_path / "out.fa"
    work = tmp_path / "work"
    assert driver.main([str(reads), "-o", str(out), "--workdir", str(work), "--polisher", POL, "--scrub", "--platform", "ont", "-t", "2"]) == 0
    lines = open(log).read().strip().split("\n")
    scrubbed = str(work / "reads.scrubbed.fq")
    tmp1 = str(work / "reads.corrected.tmp1.fa")
    assert lines[0] == f"minimap2 -x ava-ont -g 500 -t 2 {reads} {reads}"                          # scripts/vechat:198
    assert lines[1] == f"yacrd -i {work / 'scrub.paf'} -o {work / 'report.yacrd'} -c 4 -n 0.4 scrubb -i {reads} -o {scrubbed}"   # :199
    rest = sorted(lines[2:])                                 # the members of a pipeline start together: their log order is not fixed
    assert rest == sorted([f"minimap2 -x ava-ont --dual=yes {scrubbed} {scrubbed} -t 2",           # round 1, :36
                           "fpa drop --same-name --internalmatch -",
                           f"minimap2 -cx ava-ont --dual=yes {tmp1} {tmp1} -t 2",                  # round 2: base-level, :47
                           "fpa drop --same-name --internalmatch -"])
    calls = [json.loads(l) for l in open(tmp_path / "polisher.log")]
    assert calls[0][-1] == scrubbed and calls[0][-3] == scrubbed                                   # the scrubbed reads are queries and targets
    assert [l[1:] for l in open(out).read().split("\n")[0::2] if l] == [n for n, _ in recs]


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
