"""Test stand-in for `minimap2 -x ava-* | awk | fpa` (tests/test_driver.py): all-vs-all overlaps of simulated reads whose
names carry their origin, `<id>_<genome start>_<genome end>_<strand>`, written as plain PAF (no CIGAR: the polisher aligns
them on the device, like the reference aligns minimap2's PAF with edlib).  Coordinates are interpolated from the genome
interval, so they are a few bases off at the ends -- like a seed-chain overlapper's.

  python stub_overlapper.py <targets> <reads> <out.paf> [min overlap]"""
import gzip
import sys


def records(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        lines = [l.rstrip("\n") for l in f]
    per = 4 if lines and lines[0].startswith("@") else 2
    return [(lines[i][1:].split()[0], len(lines[i + 1])) for i in range(0, len(lines) - 1, per)]


def origin(name):
    p = name.split("_")
    return int(p[-3]), int(p[-2]), p[-1][:1]          # fragment correction appends an 'r' to the name per round (src/polisher.cpp:525)


def span(lo, hi, g0, g1, length, strand):
    f0, f1 = (g0 - lo) / (hi - lo), (g1 - lo) / (hi - lo)
    a, b = int(round(f0 * length)), int(round(f1 * length))
    return (a, b) if strand == "+" else (length - b, length - a)


def main():
    targets, reads, out = sys.argv[1:4]
    min_ovl = int(sys.argv[4]) if len(sys.argv) > 4 else 500
    T, Q = records(targets), records(reads)
    with open(out, "w") as fw:
        for qn, ql in Q:
            q0, q1, qs = origin(qn)
            for tn, tl in T:
                if tn == qn:
                    continue
                t0, t1, ts = origin(tn)
                g0, g1 = max(q0, t0), min(q1, t1)
                if g1 - g0 < min_ovl:
                    continue
                qa, qb = span(q0, q1, g0, g1, ql, qs)
                ta, tb = span(t0, t1, g0, g1, tl, ts)
                if qb - qa < min_ovl or tb - ta < min_ovl:
                    continue
                n = min(qb - qa, tb - ta)
                fw.write("\t".join(map(str, [qn, ql, qa, qb, "+" if qs == ts else "-", tn, tl, ta, tb, int(0.85 * n), max(qb - qa, tb - ta), 60])) + "\n")


if __name__ == "__main__":
    main()
