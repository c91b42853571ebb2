#!/usr/bin/env python3
"""Per-launch counters of one kernel from rocprofv3 PMC passes -> the JSON record bench.py reads (profiles/rNN_*_counters.json).

usage: collect_counters.py <kernel-substring> <grid_x> <batch> <out.json> <db> [<db> ...]
Every db is one `rocprofv3 --pmc ...` pass (separate passes, --kernel-trace only); counters are looked up in whichever pass has them.
  * HBM traffic = 2 * FETCH_SIZE + WRITE_SIZE (KB): on gfx950 FETCH_SIZE tallies 128-byte requests as 64 bytes
    (/opt/skills/guides/MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as reported.
  * VALU lane-operations = SQ_INSTS_VALU * 64 (wave64 instructions; the kernels run full waves).
  * SIMD VALU busy = SQ_ACTIVE_INST_VALU / (4 * SQ_BUSY_CYCLES-equivalent): reported as SQ_ACTIVE_INST_VALU (quad-cycles summed over
    waves) divided by (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 = cycles of the launch) * 1024 SIMDs / 4.
"""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


KSMAC_SOURCES = ("ksmac_kernels.hpp", "dev_math_f64.hpp", "dev_math.hpp", "ntt_kernels.hpp", "troyn_ksmac2.hip", "launch.hpp", "troyn.hip")      # = bench.py KSMAC_SOURCES


def kernel_sources_sha(names=KSMAC_SOURCES):
    """stamp of the kernel sources this record was measured on; bench.py drops a record whose stamp is not the current sources'"""
    h = hashlib.sha256()
    for nm in names:
        with open(os.path.join(ROOT, "troy-nova_amd", "csrc", nm), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def lookup(dbs, counter, kernel, grid):
    for db in dbs:
        con = sqlite3.connect(db)
        try:
            row = con.execute("select avg(value), count(*), avg(duration) from counters_collection where counter_name=? and kernel_name like ? and grid_size_x=?",
                              (counter, "%" + kernel + "%", grid)).fetchone()
        except sqlite3.Error:
            continue
        if row and row[1]:
            return row
    return None


def main():
    kernel, grid, batch, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dbs = sys.argv[5:]
    res = {"kernel": kernel, "grid_x": grid, "batch": batch, "kernel_src_sha": kernel_sources_sha()}
    get = lambda c: lookup(dbs, c, kernel, grid)
    f, w = get("FETCH_SIZE"), get("WRITE_SIZE")
    if f and w:
        res.update({"FETCH_SIZE_KB_raw": f[0], "WRITE_SIZE_KB": w[0], "fetch_correction": 2.0,
                    "traffic_bytes_per_launch": (2.0 * f[0] + w[0]) * 1024.0,
                    "profiled_duration_us": {"fetch_pass": f[2] / 1e3, "write_pass": w[2] / 1e3}})
    iv, av, wc, ga = get("SQ_INSTS_VALU"), get("SQ_ACTIVE_INST_VALU"), get("SQ_WAVE_CYCLES"), get("GRBM_GUI_ACTIVE")
    if iv:
        res["SQ_INSTS_VALU"] = iv[0]
        res["valu_lane_ops_per_launch"] = iv[0] * 64.0
    for name, row in (("SQ_ACTIVE_INST_VALU", av), ("SQ_WAVE_CYCLES", wc), ("SQ_WAIT_ANY", get("SQ_WAIT_ANY")), ("SQ_WAIT_INST_ANY", get("SQ_WAIT_INST_ANY")),
                      ("SQ_ACTIVE_INST_ANY", get("SQ_ACTIVE_INST_ANY")), ("SQ_ACTIVE_INST_LDS", get("SQ_ACTIVE_INST_LDS")), ("SQ_INSTS_LDS", get("SQ_INSTS_LDS")),
                      ("SQ_INSTS_VMEM_RD", get("SQ_INSTS_VMEM_RD")), ("GRBM_GUI_ACTIVE", ga), ("TCC_HIT_sum", get("TCC_HIT_sum")), ("TCC_MISS_sum", get("TCC_MISS_sum"))):
        if row:
            res[name] = row[0]
    if av and ga:
        cycles = ga[0] / 8.0                       # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        res["launch_cycles"] = cycles
        res["effective_clock_GHz"] = cycles / (ga[2] / 1e3) / 1e3
        res["simd_valu_busy"] = round(av[0] * 4.0 / (cycles * 1024.0), 4)     # quad-cycles -> cycles, 1024 SIMDs
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
