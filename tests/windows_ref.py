"""TEST INFRASTRUCTURE: an independent pure-Python restatement of racon's window assembly, used to
cross-check vechat_amd/csrc/vc_windows.cpp (the reference's own code for this layer cannot be built
here, so neither side is pinned to it; two separately written restatements agreeing is the check).

  breaking_points   <- src/overlap.cpp:222-292
  build_windows     <- src/polisher.cpp:389-462, src/window.cpp:47-72, src/sequence.cpp:50-83
  stitch            <- src/polisher.cpp:520-547
"""
import re

_COMP = {65: 84, 84: 65, 67: 71, 71: 67}


def revcomp(s):
    return bytes(_COMP.get(c, c) for c in reversed(s))


def breaking_points(cigar, strand, q_begin, q_end, q_length, t_begin, t_end, W):
    ends = [i - 1 for i in range(0, t_end, W) if i > t_begin] + [t_end - 1]
    out, w = [], 0
    first, last, have = (0, 0), (0, 0), False
    q = (q_length - q_end if strand else q_begin) - 1
    t = t_begin - 1

    def close():
        nonlocal have, w
        if have:
            out.append(first)
            out.append(last)
        have = False
        w += 1

    for num, op in re.findall(r"(\d+)([MIDNSHP=X])", cigar):
        n = int(num)
        if op in "M=X":
            for _ in range(n):
                q += 1
                t += 1
                if not have:
                    have, first = True, (t, q)
                last = (t + 1, q + 1)
                if w < len(ends) and t == ends[w]:
                    close()
        elif op == "I":
            q += n
        elif op in "DN":
            for _ in range(n):
                t += 1
                if w < len(ends) and t == ends[w]:
                    close()
    return out


def build_windows(seqs, n_targets, overlaps, W, qthr, rank_layers):
    """seqs: [(name, data, qual|None)], overlaps: [(q_id, t_id, strand, qb, qe, ql, tb, te, cigar)].
    -> (windows, coverage); window = dict(target, rank, backbone, backbone_quality, fasta, layers=[(seq, qual|None, b, e)] in rank order)."""
    wins, first = [], [0]
    for t in range(n_targets):
        name, data, qual = seqs[t]
        k = 0
        for j in range(0, len(data), W):
            L = min(j + W, len(data)) - j
            if qual is None:
                bq, fasta = b"!" * L, L == W
            else:
                bq = qual[j:j + L]
                fasta = (j + L == len(qual)) and bq == b"!" * L
            wins.append(dict(target=t, rank=k, backbone=data[j:j + L], backbone_quality=bq, fasta=fasta, layers=[]))
            k += 1
        first.append(first[-1] + k)
    cov = [0] * n_targets
    for q_id, t_id, strand, qb, qe, ql, tb, te, cigar in overlaps:
        cov[t_id] += 1
        name, data, qual = seqs[q_id]
        d = revcomp(data) if strand else data
        ql_ = None if qual is None else (qual[::-1] if strand else qual)
        bp = breaking_points(cigar, strand, qb, qe, ql, tb, te, W)
        for j in range(0, len(bp) - 1, 2):
            (t0, q0), (t1, q1) = bp[j], bp[j + 1]
            if q1 - q0 < 0.02 * W:
                continue
            if ql_ is not None:
                avg = 0.0
                for c in ql_[q0:q1]:
                    avg += c - 33
                avg /= q1 - q0
                if avg < qthr:
                    continue
            wid = first[t_id] + t0 // W
            ws = (t0 // W) * W
            b, e = t0 - ws, t1 - ws - 1
            if q1 == q0 or b == e:
                continue
            L = len(wins[wid]["backbone"])
            assert b < e and b <= L and e <= L
            wins[wid]["layers"].append((d[q0:q1], None if ql_ is None else ql_[q0:q1], b, e))
    for w in wins:
        order = rank_layers([0] + [l[2] for l in w["layers"]])
        w["order"] = order
    return wins, cov


def stitch(wins, cov, names, consensus, polished, drop_unpolished=True, fragment=True):
    out, data, npol = [], b"", 0
    for i, w in enumerate(wins):
        npol += 1 if polished[i] else 0
        data += consensus[i]
        if i == len(wins) - 1 or wins[i + 1]["rank"] == 0:
            ratio = npol / float(w["rank"] + 1)
            if not drop_unpolished or ratio > 0:
                out.append((names[w["target"]] + ("r" if fragment else "") + " LN:i:%d RC:i:%d XC:f:%f" % (len(data), cov[w["target"]], ratio), data))
            data, npol = b"", 0
    return out
