#!/usr/bin/env python3
"""Opcode-class histogram of k_fwd's row loop, from the device ISA of the build (roofline.peak_mix in bench.py).

  python tools/valu_mix.py [out.json]      (build container: needs hipcc; ~30 s)

Compiles vc_api.hip for the device only (-S, the flags of __graft_entry__.build(), width classes 8 / 10 = config C), finds the
dominant instantiation k_fwd_dt<8, 10, 6, true>, takes its two row loops (one per width class: the region from the
loop's header label to its last backward branch, inside the 64-row block loop) and classifies every vector-ALU instruction:
  pk16_max / pk16_add   v_pk_max_i16 / v_pk_add_u16 ... (the DP itself)     perm / alignbit   v_perm_b32 / v_alignbit_b32 (cell shift, row packing)
  dpp    any instruction with a DPP modifier (lane scan, shift)             lane   v_readlane / v_writelane / v_readfirstlane
  i32_misc   everything else (moves, 32-bit max / add / logic, address arithmetic, compares, selects)
Two histograms:  `static` = every instruction of the loops once;  `representative_row` = the instructions on the CHEAPEST way round
each loop (shortest path through the loop's control-flow graph by instruction count, header to back edge): the row whose only
predecessor is the row above and that has nothing extra to do -- by far the commonest row; the rarer rows add predecessor merges
(packed max + LDS reads), LDS ring writes and end-cell bookkeeping on top.  The second is what bench.py weights the class rates
with; the PMC count of VALU instructions per DP row scales it to the job."""
import json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_Z8k_fwd_dtILi8ELi10ELi6ELb1EEv9VcFwdArgs"        # the build phase's forward kernel (round 6: doubly tilted rows, vc_fwd_dt.h)


def cls(op, rest):
    """class of a vector-ALU instruction = the calibration test of tools/valu_peak.hip that prices it (class_<name>)"""
    if "dpp" in op or "row_shr" in rest or "row_bcast" in rest or "wave_shr" in rest or "quad_perm" in rest:
        return "dpp"
    if op.startswith("v_pk_max") or op.startswith("v_pk_min"):
        return "pk16_max"
    if op.startswith("v_pk_"):
        return "pk16_add"
    base = op.split("_e")[0]
    if base == "v_perm_b32":
        return "perm"
    if base in ("v_alignbit_b32", "v_alignbyte_b32"):
        return "alignbit"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    return "i32_misc"


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r6_valu_mix.json")
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "vc_api.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-I", os.path.join(ROOT, "include"),
                               "-mllvm", "-structurizecfg-skip-uniform-regions", "-DVC_FAST_BUILD", "--cuda-device-only", "-S",
                               os.path.join(ROOT, "vechat_amd", "csrc", "vc_api.hip"), "-o", asm], stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = [l.split(";")[0].rstrip() for l in lines[start:end]]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    back = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), 1 << 30) < i:
            back.append((labels[m.group(1)], i))
    # loops by header; the row loop of a width class = the largest loop nested inside the largest loop of that body (the 64-row block loop)
    span = {}
    for a, b in back:
        span[a] = max(span.get(a, 0), b)
    loops = sorted(span.items(), key=lambda x: x[0] - x[1])          # by size, largest first
    outer = []
    for a, b in loops:
        if not any(oa <= a and b <= ob for oa, ob in outer):
            outer.append((a, b))
        if len(outer) == 2:
            break
    # the row loop of a body: the largest loop inside it whose header fetches the row's record word (v_readlane_b32 of the prefetched records) --
    # round 6's kernel has one more level between the 64-row block loop and the rows (the band block of 8 rows)
    def first_op(a):              # (the loop is rotated: its back edges go to the latch block that sits in front of the header)
        for i in range(a, min(a + 12, len(body))):
            m = re.match(r"\s+([a-z]\S+)", body[i])
            if m and m.group(1).startswith("v_readlane"):
                return m.group(1)
        return ""
    rows = []
    for oa, ob in outer:
        inner = [(a, b) for a, b in loops if oa < a and b <= ob and first_op(a).startswith("v_readlane")]
        rows.append(max(inner, key=lambda x: x[1] - x[0]))
    static, rep = {}, {}
    n_blocks = 0
    paths = []
    for a, b in rows:
        # basic blocks of the loop: [first line, last line], instructions, successors
        starts = sorted({a} | {i for i in range(a, b + 1) if re.match(r"^\.LBB", body[i])} |
                        {i + 1 for i in range(a, b) if re.match(r"\s+s_c?branch", body[i]) or re.match(r"\s+s_setpc", body[i])})
        starts = [x for x in starts if x <= b]
        blocks = []
        for k, st in enumerate(starts):
            en = (starts[k + 1] - 1) if k + 1 < len(starts) else b
            ins = [(m.group(1), m.group(2)) for i in range(st, en + 1) for m in [re.match(r"\s+([a-z]\S+)\s*(.*)", body[i])] if m]
            blocks.append({"st": st, "en": en, "ins": ins})
        at = {blk["st"]: k for k, blk in enumerate(blocks)}
        lab_blk = {}
        for k, blk in enumerate(blocks):
            m = re.match(r"^(\.LBB\d+_\d+):", body[blk["st"]])
            if m:
                lab_blk[m.group(1)] = k
        for k, blk in enumerate(blocks):
            n_blocks += 1
            for op, rest in blk["ins"]:
                if op.startswith("v_"):
                    static[cls(op, rest)] = static.get(cls(op, rest), 0) + 1
            succ, latch = [], False
            last = blk["ins"][-1] if blk["ins"] else ("", "")
            m = re.match(r"(\.LBB\d+_\d+)", last[1]) if last[0].startswith(("s_branch", "s_cbranch")) else None
            if m:
                if m.group(1) in lab_blk:
                    if lab_blk[m.group(1)] == 0:
                        latch = True                                   # back to the header: the row is done
                    else:
                        succ.append(lab_blk[m.group(1)])
            if not last[0].startswith("s_branch") and k + 1 < len(blocks):
                succ.append(k + 1)
            blk["succ"], blk["latch"] = succ, latch
        # the cheapest way round the loop (by instruction count) THAT RUNS THE DP: header -> the block with the last step of the lane scan
        # (row_bcast:31) -> back to the header.  That is the row whose only predecessor is the row above, with nothing extra to do.  (The
        # rotated loop also has trivial ways round that skip the row body: exits and error arms.)
        import heapq

        def shortest(src, is_target):
            dist, prev = {src: len(blocks[src]["ins"])}, {}
            pq = [(dist[src], src)]
            while pq:
                dcur, k = heapq.heappop(pq)
                if dcur > dist.get(k, 1 << 30):
                    continue
                if is_target(k) and k != src:
                    p_, kk = [], k
                    while kk is not None:
                        p_.append(kk); kk = prev.get(kk)
                    return p_[::-1]
                for t in blocks[k]["succ"]:
                    nd = dcur + len(blocks[t]["ins"])
                    if nd < dist.get(t, 1 << 30):
                        dist[t] = nd; prev[t] = k
                        heapq.heappush(pq, (nd, t))
            return None

        scan = lambda k: any("row_bcast:31" in rest for _, rest in blocks[k]["ins"])
        p1 = shortest(0, scan)
        p2 = shortest(p1[-1], lambda k: blocks[k]["latch"]) if p1 else None
        if not p1 or not p2:
            raise SystemExit("valu_mix: no path through the lane scan found in the row loop")
        path = p1 + p2[1:]
        h = {}
        for k in path:
            for op, rest in blocks[k]["ins"]:
                if op.startswith("v_"):
                    h[cls(op, rest)] = h.get(cls(op, rest), 0) + 1
                    rep[cls(op, rest)] = rep.get(cls(op, rest), 0) + 1
        paths.append({"blocks": len(path), "instructions": sum(len(blocks[k]["ins"]) for k in path), "valu": h})
    sys.path.insert(0, ROOT)
    from bench import kernel_hash
    tot = sum(rep.values())
    res = {"kernel": "k_fwd_dt<8, 10, 6, true>", "kernel_hash": kernel_hash(), "row_loops_lines": rows, "basic_blocks": n_blocks,
           "plain_row_paths": paths, "static": static, "representative_row": rep, "fractions": {k: v / tot for k, v in sorted(rep.items())},
           "classes": "each class is priced by the test class_<name> of tools/valu_peak.hip",
           "source": "tools/valu_mix.py (device ISA of vc_api.hip built with the product's flags, width classes 8 / 10)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["representative_row"]), json.dumps(res["static"]), "->", out)


if __name__ == "__main__":
    main()
