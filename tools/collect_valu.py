#!/usr/bin/env python3
"""Vector-ALU issue accounting of EVERY kernel of a profiled workload, from one rocprofv3 SQ pass + one GRBM pass -> the JSON record bench.py reads
(profiles/rNN_<tag>_valu.json): per kernel the wave-instruction count, its lane-operations (x 64), the SIMD-busy fraction
(SQ_ACTIVE_INST_VALU x 4 cycles / (launch cycles x 1024 SIMDs)) and its share of the workload's kernel time; and the workload totals against the
chip's nominal issue rate (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-operations/s: every VALU instruction of gfx950 -- v_fma_f64,
v_mad_u64_u32, v_mul_lo/hi_u32, v_add -- occupies its SIMD for 4 cycles per wave64).
usage: collect_valu.py <out.json> <SQ db> <GRBM db> [--units N --unit-name NAME] [--pass "SUBSTR[@GRID]=COUNT" ... | --base-calls N]
  --units: work items one pass of the workload processes (e.g. 1024 ciphertext pairs), used for the per-pass / per-unit totals.
  --pass:  the kernels of ONE pass and how often each is launched in it (a profiled run may also hold other launches of the same process:
           parity checks, the three-call comparison); --base-calls N instead counts every kernel calls / N times."""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOMINAL_LANE_OPS = 256 * 4 * 16 * 2.4e9


def all_sources_sha():
    """stamp of every kernel source (bench.py computes the same value and drops a record of other sources)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "troy-nova_amd", "csrc")
    for nm in sorted(os.listdir(d)):
        if nm.endswith((".hip", ".hpp", ".inl")):
            with open(os.path.join(d, nm), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def main():
    args = sys.argv[1:]
    out, sq, grbm = args[0], args[1], args[2]
    units, unit_name, only, passes, base_calls = 0, "unit", [], [], 0
    i = 3
    while i < len(args):
        if args[i] == "--units": units = int(args[i + 1]); i += 2
        elif args[i] == "--unit-name": unit_name = args[i + 1]; i += 2
        elif args[i] == "--only": only.append(args[i + 1]); i += 2
        elif args[i] == "--pass": passes.append(args[i + 1]); i += 2
        elif args[i] == "--base-calls": base_calls = int(args[i + 1]); i += 2
        else: i += 1
    con = sqlite3.connect(sq)
    rows = con.execute("select kernel_name, grid_size_x, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                       "group by kernel_name, grid_size_x, counter_name").fetchall()
    kern = {}
    for name, gx, cname, cnt, val, dur in rows:
        k = kern.setdefault((name, gx), {"calls": cnt, "avg_us": dur / 1e3})
        k[cname] = val
    cycles = {}
    for name, gx, val, dur in sqlite3.connect(grbm).execute("select kernel_name, grid_size_x, avg(value), avg(duration) from counters_collection "
                                                           "where counter_name='GRBM_GUI_ACTIVE' group by kernel_name, grid_size_x").fetchall():
        cycles[(name, gx)] = (val / 8.0, dur / 1e3)       # summed over the 8 XCDs
    total_time = sum(k["avg_us"] * k["calls"] for (name, gx), k in kern.items() if name.startswith(("void troyn", "troyn")))
    recs = []
    for (name, gx), k in sorted(kern.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls"]):
        if not name.startswith(("void troyn", "troyn")) or "SQ_INSTS_VALU" not in k:
            continue
        if only and not any(s in name for s in only):
            continue
        r = {"kernel": name.replace("unsigned long long", "u64").replace("unsigned int", "u32")[:150], "grid_x": gx, "calls": k["calls"], "avg_us": round(k["avg_us"], 2),
             "time_share": round(k["avg_us"] * k["calls"] / total_time, 4), "valu_wave_insts": k["SQ_INSTS_VALU"], "valu_lane_ops": k["SQ_INSTS_VALU"] * 64.0}
        if (name, gx) in cycles and "SQ_ACTIVE_INST_VALU" in k:
            cyc, dur_us = cycles[(name, gx)]
            r["effective_clock_GHz"] = round(cyc / dur_us / 1e3, 3)
            r["simd_valu_busy"] = round(k["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0), 4)
        r["nominal_us_at_full_issue"] = round(r["valu_lane_ops"] / NOMINAL_LANE_OPS * 1e6, 2)
        if r["time_share"] >= 0.005:
            recs.append(r)
    res = {"sources_sha": all_sources_sha(), "nominal_lane_ops_per_s": NOMINAL_LANE_OPS, "kernels": recs}
    if units:
        # totals of one pass over `units` work items
        def count(r):
            if passes:
                for spec in passes:
                    sel, cnt = spec.rsplit("=", 1)
                    sub, _, grid = sel.partition("@")
                    if sub in r["kernel"] and (not grid or int(grid) == r["grid_x"]):
                        return float(cnt)
                return 0.0
            return round(r["calls"] / base_calls) if base_calls else 1.0
        tot_ops = sum(r["valu_lane_ops"] * count(r) for r in recs)
        tot_us = sum(r["avg_us"] * count(r) for r in recs)
        for r in recs:
            r["launches_per_pass"] = count(r)
        res["per_pass"] = {"units": units, "unit": unit_name, "valu_lane_ops": tot_ops, "nominal_ms_at_full_issue": round(tot_ops / NOMINAL_LANE_OPS * 1e3, 4),
                           "profiled_kernel_ms": round(tot_us / 1e3, 4), "issue_frac": round(tot_ops / NOMINAL_LANE_OPS / (tot_us * 1e-6), 4),
                           "valu_lane_ops_per_unit": tot_ops / units}
        res["kernels"] = [r for r in recs if r["launches_per_pass"] > 0] + [r for r in recs if r["launches_per_pass"] == 0]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res.get("per_pass", {})))


if __name__ == "__main__":
    main()
