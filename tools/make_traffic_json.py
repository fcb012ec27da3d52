"""Build profiles/r6_hbm_traffic.json and profiles/r6_valu_peak_final.json from the PMC passes of tools/gpu_round.sh and tools/valu_peak.bin.

  python tools/make_traffic_json.py <dir with pmc1..4 counter csv> <gpu_scale stats json> <valu_peak output> <out dir> [NAME=<pmc dir>:<stats json> ...]

NAME=...: one SQ_INSTS_VALU pass over another workload shape (configs E, W): its VALU wave-instructions per window go into
valu_insts_per_window_by_config (bench.py: roofline.per_config).

FETCH_SIZE is doubled (gfx950: 128-B requests counted as 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported.
Both are in KB.  The kernel hash ties the file to the sources it was measured on (bench.py refuses any other)."""
import csv, collections, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pmc_dir, stats_json, valu_out, out_dir = sys.argv[1:5]
extra = sys.argv[5:]


from bench import kernel_hash  # noqa: E402  (the same identity bench.py checks)


agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.Counter()
for f in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        k = k.split("<")[0]
        if k == "k_fwd_dt":
            k = "k_fwd"                     # the doubly tilted forward kernel (round 6) counts as the forward pass
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, r["Counter_Name"])] += 1
st = json.load(open(stats_json))            # summed over the repetitions the PMC command ran
cells, rows, moves = st["cells"], st["dp_rows"], st["trace_steps"]
out = {"source": "rocprofv3 --kernel-trace --pmc <group>, one counter group per pass (tools/pmc_run.sh), command: " + st["command"],
       "kernel_hash": kernel_hash(), "cells": cells, "dp_rows": rows, "backtrack_moves": moves,
       "fetch_correction": 2.0, "units": "FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes",
       "kernels": {}}
for k, d in sorted(agg.items()):
    e = {c: v for c, v in d.items()}
    e["dispatches"] = max(n for (kk, _), n in disp.items() if kk == k)
    if "FETCH_SIZE" in d:
        e["hbm_read_bytes"] = d["FETCH_SIZE"] * 1024 * 2.0
    if "WRITE_SIZE" in d:
        e["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
    out["kernels"][k] = e
# whole job: VALU wave-instructions of every kernel of the command, per window (bench.py prices a step against the issue peak)
tot_valu = sum(d.get("SQ_INSTS_VALU", 0.0) for d in agg.values())
if st.get("windows") and tot_valu:
    out["windows"] = st["windows"]
    out["valu_insts_per_window_all_kernels"] = tot_valu / st["windows"]
    out["valu_insts_per_window_by_kernel"] = {k: d.get("SQ_INSTS_VALU", 0.0) / st["windows"] for k, d in sorted(agg.items()) if d.get("SQ_INSTS_VALU")}
    # every instruction class the SQ counts, all kernels, per window (bench.py: roofline.instruction_issue)
    classes = ("VALU", "SALU", "BRANCH", "SMEM", "LDS", "VMEM_RD", "VMEM_WR")
    out["insts_per_window_all_kernels"] = {c: sum(d.get("SQ_INSTS_" + c, 0.0) for d in agg.values()) / st["windows"] for c in classes
                                           if any("SQ_INSTS_" + c in d for d in agg.values())}
f = out["kernels"].get("k_fwd", {})
if f:
    out["bytes_per_cell_written"] = f.get("hbm_write_bytes", 0) / cells
    out["bytes_per_cell_read"] = f.get("hbm_read_bytes", 0) / cells
    out["bytes_per_cell"] = out["bytes_per_cell_written"] + out["bytes_per_cell_read"]
    out["instructions_per_dp_row"] = {n: f.get("SQ_INSTS_" + n, 0) / rows for n in ("VALU", "SALU", "LDS", "VMEM_WR", "VMEM_RD")}
    if "SQ_INSTS_BRANCH" in f:
        out["instructions_per_dp_row"]["BRANCH"] = f["SQ_INSTS_BRANCH"] / rows
t = out["kernels"].get("k_tracew", {})
if t and moves:
    out["k_tracew_bytes_fetched_per_move"] = t.get("hbm_read_bytes", 0) / moves
by_cfg = {}
for spec in extra:
    name, rest = spec.split("=", 1)
    d2, sj = rest.split(":", 1)
    try:
        tot2 = 0.0
        for f in glob.glob(os.path.join(d2, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == "SQ_INSTS_VALU":
                    tot2 += float(r["Counter_Value"])
        st2 = json.load(open(sj))
        if tot2 and st2.get("windows"):
            by_cfg[name] = tot2 / st2["windows"]
            out.setdefault("by_config_commands", {})[name] = st2["command"]
    except Exception as e:
        print("config", name, "skipped:", repr(e))
out["valu_insts_per_window_by_config"] = by_cfg
os.makedirs(out_dir, exist_ok=True)
json.dump(out, open(os.path.join(out_dir, "r6_hbm_traffic.json"), "w"), indent=1)
print("bytes/cell", out.get("bytes_per_cell"), "insts/row", out.get("instructions_per_dp_row"), "tracew B/move", out.get("k_tracew_bytes_fetched_per_move"))

tests = [json.loads(l) for l in open(valu_out) if l.startswith("{")]
dev = tests[0]
ind = [x for x in tests[1:] if x["test"] == "pk_i16_independent"]
best = max(ind, key=lambda x: x["inst_per_us_per_simd"])
json.dump({"source": "tools/valu_peak.bin on the bench box", "device": dev, "tests": tests[1:],
           "peak_wave_insts_per_us_per_simd": best["inst_per_us_per_simd"],
           "note": f"best issue rate of independent v_pk_max_i16 / v_pk_add_i16 chains ({best['waves_per_simd']} waves per SIMD)"},
          open(os.path.join(out_dir, "r6_valu_peak_final.json"), "w"), indent=1)
print("valu peak", best["inst_per_us_per_simd"], "wave insts / us / SIMD")
