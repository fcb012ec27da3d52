"""The VeChat driver over the MI355X polisher: two rounds of overlap -> correct, the way the reference's wrapper runs them
(scripts/vechat:17-97 run_error_correction, :300-397 the round loop with and without --split).

  python -m vechat_amd.driver reads.fastq -o reads.corrected.fa [--platform pb|ont] [--split] [--gpus N]

Round 1 (haplotype-aware, variation graph): overlaps of all reads against the target chunk, then
`polish -f -p -d D -s S`.  Round 2 (linear consensus of the round-1 output against itself): base-level overlaps filtered
by --min-ovlplen-cns / --min-identity-cns, then `polish -f [-u]`.  --linear runs round-2-style correction once.  The
corrected FASTA of round i is the input of round i+1 (reads.corrected.tmp<i>.fa), the last one is moved to --outfile and
the temporaries are removed (scripts/vechat:371-397).

The external tools are commands, not libraries, exactly as in the reference (minimap2 | awk | fpa, yacrd): this module
only builds their command lines.  They are pluggable templates so that another overlapper -- or the stub of
tests/test_driver.py -- can stand in:
  --overlapper-r1 / --overlapper-r2   fields {platform} {targets} {reads} {threads} {out} {min_ovlplen} {min_identity}
  --polisher                          the command that replaces build/bin/vechat_racon; default: this package's
                                      `python -m vechat_amd.polish` (one process per GPU under torch.distributed.run when --gpus > 1)
Nothing here computes on the CPU what the reference computes in vechat_racon; without a GPU the polisher command fails.
"""
import argparse
import gzip
import os
import re
import shlex
import shutil
import subprocess
import sys

# scripts/vechat:36-38 (default), :41-43 (--base), :47-49 (round 2)
OVERLAPPER_R1 = ("minimap2 -x ava-{platform} --dual=yes {targets} {reads} -t {threads} 2>/dev/null|awk '$11>=500'|"
                 "fpa drop --same-name --internalmatch  - >{out}")
OVERLAPPER_R1_BASE = ("minimap2 -cx ava-{platform} --dual=yes {targets} {reads} -t {threads} 2>/dev/null|"
                      "awk '$11>=500 && $10/$11>={min_identity}'|cut -f 1-12|fpa drop --same-name --internalmatch  - >{out}")
OVERLAPPER_R2 = ("minimap2 -cx ava-{platform} --dual=yes {targets} {reads} -t {threads} 2>/dev/null|"
                 "awk '$11>={min_ovlplen} && $10/$11>={min_identity}'|cut -f 1-12|fpa drop --same-name --internalmatch  - >{out}")


def fq_or_fa(path):
    """scripts/vechat:173-187: the first byte decides."""
    with (gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "r")) as fr:
        c = fr.readline()[:1]
    if c == ">":
        return "fa"
    if c == "@":
        return "fq"
    raise ValueError(f"invalid input file, must be FASTA/FASTQ format: {path}")


COMMAND_LOG = None       # tests set this to a list: every external command the driver issues is appended (tests/test_driver.py)


def _sh(cmd, cwd):
    print(f"[vechat_amd.driver] {cmd}", file=sys.stderr)
    if COMMAND_LOG is not None:
        COMMAND_LOG.append(cmd)
    env = dict(os.environ)                                    # the default polisher is this package: make it importable from the work directory
    pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = pkg_parent + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    # pipefail: a missing or failing minimap2 must not end as an empty overlap.paf that only shows up rounds later
    rc = subprocess.call(["/bin/bash", "-o", "pipefail", "-c", cmd], cwd=cwd, env=env)
    if rc != 0:
        raise RuntimeError(f"command failed ({rc}): {cmd}")


def split_lines(path, lines_per_chunk, suffix, workdir, prefix="reads_chunk"):
    """`split -l N -d --additional-suffix .<suffix> <path> reads_chunk` (scripts/vechat:313-314): numbered chunks of N lines."""
    out, k, buf = [], 0, []

    def flush():
        nonlocal k, buf
        if buf:
            name = os.path.join(workdir, f"{prefix}{k:02d}.{suffix}")
            with open(name, "w") as fw:
                fw.writelines(buf)
            out.append(name)
            k += 1
            buf = []

    with (gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "r")) as fr:
        for line in fr:
            buf.append(line)
            if len(buf) == lines_per_chunk:
                flush()
    flush()
    return out


def _open_text(path):
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "r")


def extract_sub_sequences(sequences, overlap, chunk_targets, workdir):
    """scripts/vechat:99-171: with --split only the reads that share an overlap with a target of the chunk are handed to the
    polisher (the unwrapped 4-line FASTQ / 2-line FASTA layout the reference assumes; gzip input is read transparently,
    where the reference's plain open() would fail)."""
    mode = fq_or_fa(chunk_targets)
    per = 4 if mode == "fq" else 2
    names = set()
    with _open_text(chunk_targets) as fr:
        for i, line in enumerate(fr):
            if i % per == 0:
                names.add(line[1:].rstrip().split()[0])
    wanted = set()
    with open(overlap) as fr:
        for line in fr:
            a = line.split()
            if len(a) > 5 and (a[0] in names or a[5] in names):
                wanted.add(a[0]); wanted.add(a[5])
    mode_q = fq_or_fa(sequences)
    per_q = 4 if mode_q == "fq" else 2
    out = os.path.join(workdir, f"query_sequences.tmp.{mode_q}")
    keep = False
    with _open_text(sequences) as fr, open(out, "w") as fw:
        for i, line in enumerate(fr):
            if i % per_q == 0:
                keep = line[1:].rstrip().split()[0] in wanted
            if keep:
                fw.write(line)
    return out


_FLOAT = re.compile(r"[0-9]+(\.[0-9]*)?([eE][-+]?[0-9]+)?|\.[0-9]+([eE][-+]?[0-9]+)?")
_INT = re.compile(r"[0-9]+")


_SHARED = {}             # --in-process: the polisher's context, kept from one invocation to the next (vechat_amd.polish.main(shared=...))


def close_shared():
    ctx = _SHARED.pop("ctx", None)
    if ctx is not None:
        ctx.close()
    _SHARED.clear()


def polisher_command(a):
    if a.polisher:
        return a.polisher
    if a.gpus > 1:
        return (f"{shlex.quote(sys.executable)} -m torch.distributed.run --nnodes=1 --nproc-per-node {a.gpus} --master-addr 127.0.0.1 "
                f"--master-port {a.master_port} -m vechat_amd.polish")
    return f"{shlex.quote(sys.executable)} -m vechat_amd.polish"


def run_error_correction(a, sequences, chunk_targets, corrected_file, iteration, workdir):
    """One overlap + correct pass over one chunk of targets (scripts/vechat:17-97)."""
    linear = a.linear or iteration == 2                      # the second iteration computes the consensus (scripts/vechat:26-27)
    overlap = os.path.join(workdir, "overlap.paf")
    fields = dict(platform=a.platform, targets=shlex.quote(chunk_targets), reads=shlex.quote(sequences), threads=a.threads,
                  out=shlex.quote(overlap))
    if iteration == 1:
        tmpl = a.overlapper_r1 or (OVERLAPPER_R1_BASE if a.base else OVERLAPPER_R1)
        cmd = tmpl.format(min_identity=a.min_identity, min_ovlplen=500, **fields)
    else:
        tmpl = a.overlapper_r2 or OVERLAPPER_R2
        cmd = tmpl.format(min_identity=a.min_identity_cns, min_ovlplen=a.min_ovlplen_cns, **fields)
    _sh(cmd, workdir)
    sub_reads = extract_sub_sequences(sequences, overlap, chunk_targets, workdir) if a.split else sequences
    pol = polisher_command(a)
    if not linear:
        print("perform variation graph based (haplotype-aware) error correction", file=sys.stderr)
        flags = f"-f -p -d {a.min_confidence} -s {a.min_support} -t {a.threads}"          # scripts/vechat:70-72
    else:
        print("perform linear sequence based error correction", file=sys.stderr)
        flags = f"-f -u -t {a.threads}" if a.include_unpolished else f"-f  -t {a.threads}"   # scripts/vechat:82-84, 91-93
    if a.cuda_banded_alignment:
        # scripts/vechat:59-66,76-80,86-89: the accelerator switches reach the polisher only together with -b (the polisher
        # here is always the accelerated one and accepts them for what they are: hints for another device)
        flags += f" -b --cudaaligner-batches {a.cudaaligner_batches} -c {a.cudapoa_batches}"
    if getattr(a, "keep_going", False) and not a.polisher:
        flags += " --keep-going"                               # (only our own polisher knows the switch)
    if getattr(a, "in_process", False) and not a.polisher and a.gpus <= 1:
        # the polisher inside this process, on ONE context that stays warm from invocation to invocation (vc_set_polish_params switches
        # the round's parameters in place): scripts/vechat:371-393 starts a process per round and per --split chunk, and each start pays
        # the device start-up and the workspaces again -- seconds that a small input consists of
        import contextlib
        from . import polish
        argv = shlex.split(flags) + [sub_reads, overlap, chunk_targets]
        print(f"[vechat_amd.driver] (in process) python -m vechat_amd.polish {' '.join(shlex.quote(x) for x in argv)} >{corrected_file}", file=sys.stderr)
        if COMMAND_LOG is not None:
            COMMAND_LOG.append(f"(in process) {flags} {sub_reads} {overlap} {chunk_targets} >{corrected_file}")
        cwd = os.getcwd()
        os.chdir(workdir)
        try:
            with open(corrected_file, "w") as fw, contextlib.redirect_stdout(fw):
                rc = polish.main(argv, shared=_SHARED)
        finally:
            os.chdir(cwd)
        if rc != 0:
            raise RuntimeError(f"polisher failed ({rc}) in process: {flags}")
    else:
        _sh(f"{pol} {flags} {shlex.quote(sub_reads)} {shlex.quote(overlap)} {shlex.quote(chunk_targets)} >{shlex.quote(corrected_file)}", workdir)
    for f in os.listdir(workdir):
        if f.startswith("query_sequences.tmp."):
            os.remove(os.path.join(workdir, f))
    return corrected_file


def scrub_reads(a, sequences, workdir):
    """scripts/vechat:189-205: yacrd on a self-overlap, platform-specific parameters."""
    suffix = fq_or_fa(sequences)
    overlap = os.path.join(workdir, "scrub.paf")
    scrubbed = os.path.join(workdir, f"reads.scrubbed.{suffix}")
    if a.platform == "pb":
        g, c = 5000, 3
    elif a.platform == "ont":
        g, c = 500, 4
    else:
        raise ValueError("Invalid platform, must be: pb or ont")
    _sh(f"minimap2 -x ava-{a.platform} -g {g} -t {a.threads} {shlex.quote(sequences)} {shlex.quote(sequences)} > {shlex.quote(overlap)}", workdir)
    _sh(f"yacrd -i {shlex.quote(overlap)} -o {shlex.quote(os.path.join(workdir, 'report.yacrd'))} -c {c} -n 0.4 scrubb -i {shlex.quote(sequences)} "
        f"-o {shlex.quote(scrubbed)}", workdir)
    os.remove(overlap)
    return scrubbed


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m vechat_amd.driver", description="Haplotype-aware error correction of noisy long reads "
                                 "(the VeChat driver, scripts/vechat) over the MI355X polisher", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument("sequences", help="FASTA/FASTQ (optionally gzip) with the reads: queries and targets of both rounds")
    ap.add_argument("-o", "--outfile", default="reads.corrected.fa")
    ap.add_argument("--platform", default="pb", help="sequencing platform: pb/ont")
    ap.add_argument("--split", action="store_true", help="split the target sequences into chunks")
    ap.add_argument("--split-size", default=1000000, type=int, help="chunk size in lines, with --split")
    ap.add_argument("--scrub", action="store_true", help="scrub chimeric reads first (minimap2 + yacrd)")
    ap.add_argument("-u", "--include-unpolished", action="store_true")
    ap.add_argument("--base", action="store_true", help="base-level alignment for the round-1 overlaps")
    # pass-through values stay the strings the user typed, as in the reference (they are only ever formatted into commands)
    ap.add_argument("--min-identity", default="0.8")
    ap.add_argument("--linear", action="store_true", help="one round of linear (racon) correction instead of the two rounds")
    ap.add_argument("-d", "--min-confidence", default="0.2")
    ap.add_argument("-s", "--min-support", default="0.2")
    ap.add_argument("--min-ovlplen-cns", default="1000")
    ap.add_argument("--min-identity-cns", default="0.99")
    ap.add_argument("-t", "--threads", default="1")
    # scripts/vechat:262-275: parsed by the reference's wrapper and never handed to vechat_racon -- accepted, same (no) effect
    for flag, long, dflt in (("-w", "--window-length", 500), ("-q", "--quality-threshold", 10.0), ("-e", "--error-threshold", 0.3),
                             ("-m", "--match", 5), ("-x", "--mismatch", -4), ("-g", "--gap", -8)):
        ap.add_argument(flag, long, default=dflt, help="accepted like the reference's wrapper accepts it: not passed on to the polisher")
    ap.add_argument("--cudaaligner-batches", default=0, help="passed on with -b (scripts/vechat:59-66)")
    ap.add_argument("-c", "--cudapoa-batches", default=0, help="passed on with -b")
    ap.add_argument("-b", "--cuda-banded-alignment", action="store_true", help="hand -b --cudaaligner-batches N -c N to the polisher")
    ap.add_argument("--gpus", default=1, type=int, help="GPUs of this node for the polisher (one process per GPU)")
    ap.add_argument("--master-port", default=29561, type=int)
    ap.add_argument("--workdir", default=".", help="where the round files live (the reference uses the current directory)")
    ap.add_argument("--overlapper-r1", default=None, help="command template replacing the minimap2|awk|fpa pipeline of round 1")
    ap.add_argument("--overlapper-r2", default=None, help="... of round 2")
    ap.add_argument("--polisher", default=None, help="command replacing `python -m vechat_amd.polish`")
    ap.add_argument("--in-process", action="store_true", help="run this package's polisher inside the driver process, on one device context that "
                    "stays warm across the rounds and --split chunks (default: a process per invocation, as the reference's wrapper does)")
    ap.add_argument("--keep-going", action="store_true", help="handed to the polisher: a window the device cannot hold keeps its backbone "
                    "(unpolished) instead of ending the run with exit status 3; VC_KEEP_GOING=1 in the environment does the same")
    a = ap.parse_args(argv)
    # The pass-through values stay the strings the user typed (byte-identical commands, tests/golden/driver_cmds.json), but
    # they go into `bash -c` command lines and an awk program: nothing but a plain number gets that far
    for name, pat in (("min_identity", _FLOAT), ("min_confidence", _FLOAT), ("min_support", _FLOAT), ("min_identity_cns", _FLOAT),
                      ("min_ovlplen_cns", _INT), ("threads", _INT), ("cudaaligner_batches", _INT), ("cudapoa_batches", _INT)):
        if not pat.fullmatch(str(getattr(a, name))):
            ap.error(f"--{name.replace('_', '-')}: {getattr(a, name)!r} is not a number")
    if a.platform not in ("pb", "ont"):
        ap.error("--platform must be pb or ont")
    a.keep_going = a.keep_going or os.environ.get("VC_KEEP_GOING") == "1"

    workdir = os.path.abspath(a.workdir)
    os.makedirs(workdir, exist_ok=True)
    iterations = 1 if a.linear else 2                         # scripts/vechat:283-286
    sequences = os.path.abspath(a.sequences)
    if a.scrub:
        print("Scrubbing reads...", file=sys.stderr)
        sequences = scrub_reads(a, sequences, workdir)
    corrected = ""
    for i in range(1, iterations + 1):
        print(f"Performing the {i} iteration for error correction...", file=sys.stderr)
        src = sequences if i == 1 else os.path.join(workdir, f"reads.corrected.tmp{i - 1}.fa")
        corrected = os.path.join(workdir, f"reads.corrected.tmp{i}.fa")
        if not a.split:
            run_error_correction(a, src, src, corrected, i, workdir)
            continue
        # scripts/vechat:300-361: round 1 chunks the input by --split-size lines; later rounds chunk the FASTA of the previous
        # round, with half as many lines when the original was FASTQ (2 lines per record instead of 4)
        suffix = fq_or_fa(sequences)
        lines = a.split_size if (i == 1 or suffix == "fa") else a.split_size // 2
        chunks = split_lines(src, lines, suffix if i == 1 else "fa", workdir)
        parts = []
        for j, chunk in enumerate(chunks, 1):
            print(f"processing chunk {j}...", file=sys.stderr)
            parts.append(run_error_correction(a, src, chunk, os.path.join(workdir, f"reads.corrected.tmp.chunk{j}.fa"), i, workdir))
        with open(corrected, "wb") as fw:
            for pth in parts:
                with open(pth, "rb") as fr:
                    shutil.copyfileobj(fr, fw)
                os.remove(pth)
        for chunk in chunks:
            os.remove(chunk)
    close_shared()
    shutil.move(corrected, a.outfile if os.path.isabs(a.outfile) else os.path.join(os.getcwd(), a.outfile))
    for f in os.listdir(workdir):                             # scripts/vechat:371-373, 395-396
        if (f.startswith("reads.corrected.tmp") and f.endswith(".fa")) or f.startswith("reads_chunk"):
            os.remove(os.path.join(workdir, f))
    return 0


if __name__ == "__main__":
    sys.exit(main())
