#!/usr/bin/env python3
"""Lint of the device ISA of a built library (run by __graft_entry__.build(); exit code 1 = a finding).

  python tools/isa_lint.py [vechat_amd/lib/libvechat_hip.so]

Why: the hot kernels contain inline assembly, and the compiler's hazard recogniser does not look inside an asm block.  Round 5 met the
consequence (NOTES.md): gfx950 wants wait states between a VMEM store of more than 64 bits and a VALU write of the store's DATA
registers; with the store inside an asm block and the next row's pack right behind it, a third of the windows came back wrong, a
different third on every run.  This walks the disassembly of every kernel of the gfx950 code object and checks, for what the asm blocks
of vc_kernels.h / vc_fwd_dt.h can get wrong:

  R1  [global|buffer|flat|scratch]_store_dwordx3 / x4: none of the next instructions within 2 wait states is a VALU instruction
      that writes one of the store's data VGPRs (s_nop N counts N + 1 wait states, any other instruction 1);
  R2  a DPP instruction does not read a VGPR that one of the two instructions in front of it wrote with a VALU instruction
      (2 wait states between a VALU write and a DPP read of the same register);
  R3  v_readlane / v_writelane with an SGPR lane select written by a VALU instruction (v_readlane, v_readfirstlane, v_cmp) within the
      last 4 wait states.

Also reports per kernel: VGPRs, SGPR / VGPR spills and scratch, from the note records; --max-warnings N lets build() fail on compiler
warnings it counted itself."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(lib, tmp):
    """the gfx950 code object inside the host library's fat binary -> path of an ELF"""
    out = os.path.join(tmp, "dev.co")
    # the fat binary is a clang offload bundle in section .hip_fatbin
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    targets = subprocess.check_output([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", "--input=" + fat]).decode().split()
    tgt = next((t for t in targets if "gfx950" in t), None)
    if not tgt:
        raise SystemExit(f"isa_lint: no gfx950 code object in {lib} (targets: {targets})")
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat, "--targets=" + tgt, "--output=" + out])
    return out


REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(tok):
    s = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            s.add(int(m.group(1)))
        else:
            s.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return s


def sregs(tok):
    s = set()
    for m in re.finditer(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]", tok):
        if m.group(1) is not None:
            s.add(int(m.group(1)))
        else:
            s.update(range(int(m.group(2)), int(m.group(3)) + 1))
    if re.search(r"\bvcc\b", tok):
        s.add(-1)
    return s


def parse(line):
    """'  v_add_u32_e32 v1, v2, v3  // 0000: ...' -> (op, [operands])"""
    line = line.split("//")[0].split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    parts = line.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return op, ops


def is_valu(op):
    return op.startswith("v_") and not op.startswith(("v_readlane", "v_readfirstlane")) or op.startswith(("v_readlane", "v_readfirstlane"))


def wait_states(op, ops):
    if op == "s_nop":
        try:
            return int(ops[0], 0) + 1
        except Exception:
            return 1
    return 1


def lint_kernel(name, insts):
    findings = []
    n = len(insts)
    for i, (op, ops) in enumerate(insts):
        # R1: wide store, then a VALU write of its data registers too soon
        if re.match(r"(global|buffer|flat|scratch)_store_dwordx[34]$", op) and ops:
            # data operand: global / flat / scratch: 2nd operand; buffer: 1st
            data = vregs(ops[0] if op.startswith("buffer") else (ops[1] if len(ops) > 1 else ""))
            ws, j = 0, i + 1
            while j < n and ws < 2:
                o2, p2 = insts[j]
                if o2.startswith("v_") and p2 and not o2.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                    if vregs(p2[0]) & data:
                        findings.append(f"R1 {name}: '{op} {', '.join(ops)}' then after {ws} wait state(s) '{o2} {', '.join(p2)}' writes its data")
                        break
                if o2.startswith(("s_branch", "s_cbranch", "s_setpc", "s_endpgm")):
                    break                                   # (a taken branch costs more than two wait states; the fall-through path is checked)
                ws += wait_states(o2, p2)
                j += 1
        # R2: DPP read right behind a VALU write
        if op.startswith("v_") and any(("row_" in o or "wave_" in o or "quad_perm" in o or "row_bcast" in o) for o in ops):
            srcs = set()
            for o in ops[1:]:
                if re.match(r"^(v\d+|v\[\d+:\d+\])", o):
                    srcs |= vregs(o.split()[0])
            # only the DPP operand (src0) is subject to the rule; being conservative costs nothing: the compiler keeps 2 wait states for all
            src0 = vregs(ops[1].split()[0]) if len(ops) > 1 else set()
            ws, j = 0, i - 1
            while j >= 0 and ws < 2:
                o2, p2 = insts[j]
                if o2.startswith(("s_branch", "s_cbranch")) or o2.endswith(":"):
                    break
                if o2.startswith("v_") and p2 and not o2.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and (vregs(p2[0]) & src0):
                    findings.append(f"R2 {name}: '{o2} {', '.join(p2)}' then after {ws} wait state(s) DPP '{op} {', '.join(ops)}' reads it")
                    break
                ws += wait_states(o2, p2)
                j -= 1
        # R3: lane select SGPR fresh from a VALU instruction
        if op.startswith(("v_readlane", "v_writelane")) and len(ops) >= 3:
            sel = sregs(ops[2])
            ws, j = 0, i - 1
            while sel and j >= 0 and ws < 4:
                o2, p2 = insts[j]
                if o2.startswith(("s_branch", "s_cbranch")):
                    break
                if o2.startswith(("v_readlane", "v_readfirstlane", "v_cmp")) and p2 and (sregs(p2[0]) & sel):
                    findings.append(f"R3 {name}: '{o2} {', '.join(p2)}' then after {ws} wait state(s) '{op} {', '.join(ops)}' selects a lane with it")
                    break
                ws += wait_states(o2, p2)
                j -= 1
    return findings


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(ROOT, "vechat_amd", "lib", "libvechat_hip.so")
    with tempfile.TemporaryDirectory() as tmp:
        co = code_object(lib, tmp)
        dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co]).decode("utf-8", "replace")
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co]).decode("utf-8", "replace")
    kernels, cur, name = {}, None, None
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1)
            cur = kernels.setdefault(name, [])
            continue
        if cur is None:
            continue
        p = parse(line)
        if p:
            cur.append(p)
    findings = []
    for k, insts in kernels.items():
        findings += lint_kernel(k, insts)
    # resource summary from the notes
    res = []
    for blk in notes.split("- .agpr_count")[1:]:
        nm = re.search(r"\.name:\s+(\S+)", blk)
        g = lambda key: (re.search(r"\." + key + r":\s+(\d+)", blk) or [None, "?"])[1]
        if nm:
            res.append((nm.group(1), g("vgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    ninst = sum(len(v) for v in kernels.values())
    print(f"isa_lint: {len(kernels)} kernels, {ninst} instructions in {os.path.basename(lib)}; {len(findings)} finding(s)")
    if "--resources" in sys.argv:
        for r in sorted(res):
            print("  %-70s vgpr %s  sgpr-spill %s  vgpr-spill %s  scratch %s  lds %s" % r)
    for f in findings[:50]:
        print("  " + f)
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
