#!/usr/bin/env python3
"""PMC counters of a rocprofv3 --pmc run of bench.py, per kernel, and the record bench.py quotes
(profiles/pmc_r05.json). The pass kernel is launched for passes on M, for passes on a row view
(far fewer bytes) and for iterations that do nothing: a launch counts as a PASS ON M when its counter reads at
least 0.6 x the largest value any launch of that kernel shows in the run.
  tools/pmc_summary.py --key m10000_csc --bytes <bytes per pass> --commit <sha> --json profiles/pmc_r05.json <db> [...]"""
import argparse
import json
import os
import sqlite3
import sys
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dbs", nargs="+")
    ap.add_argument("--key")
    ap.add_argument("--bytes", type=float, default=None)
    ap.add_argument("--commit", default="")
    ap.add_argument("--json", default=None)
    ap.add_argument("--sources-sha", default="", help="sha256 over the kernel sources the counters were measured on "
                    "(bench.py:kernel_sources_sha256): bench.py refuses a record taken on other sources")
    a = ap.parse_args()
    per = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [(value, duration)]
    for db in a.dbs:
        con = sqlite3.connect(db)
        for name, cname, val, dur in con.execute(
                "select kernel_name, counter_name, value, duration from counters_collection"):
            short = name.split("(")[0].replace("void clipper_hip::", "").replace("clipper_hip::", "")
            per[short][cname].append((val, dur))
    summary = {}
    for k in sorted(per):
        for c in sorted(per[k]):
            rows = per[k][c]
            # The passes ON M among the launches of the pass kernel (and their tails): the launches that do the most work — by
            # the counter's own value, not by duration: since round 6 a pass on M multiplies candidate 0 alone and is SHORTER
            # at m = 10k than the transition iterations whose sweeps run in one workgroup (37 us, next to no traffic).
            vmax = max(v for v, _ in rows)
            keep = [(v, d) for v, d in rows if v >= 0.6 * vmax] if (k.startswith(("k_gemv", "k_tail")) and vmax > 0) else rows
            sel = sorted(v for v, _ in keep)
            durs = sorted(d for _, d in keep)
            med = sel[len(sel) // 2]
            summary.setdefault(k, {})[c] = dict(n=len(sel), of=len(rows), median=med, mean=sum(sel) / len(sel),
                                                median_duration_us=durs[len(durs) // 2] / 1e3)
            print(f"{k[:46]:46s} {c:24s} n={len(sel):4d}/{len(rows):4d} median {med:16.1f} mean {sum(sel)/len(sel):16.1f}")
    if not (a.json and a.key):
        return
    rec = {}
    if os.path.exists(a.json):
        rec = json.load(open(a.json))
    e = rec.setdefault(a.key, {})
    e["commit"] = a.commit
    if a.sources_sha:
        e["kernel_sources_sha256"] = a.sources_sha
    if a.bytes is not None:
        e["pass_bytes_per_launch"] = a.bytes
    pk = next((k for k in summary if k.startswith("k_gemv_slices")), None) or next((k for k in summary if k.startswith("k_gemv")), None)
    if pk:
        s = summary[pk]
        if "FETCH_SIZE" in s:   # KB; a wide coalesced read is tallied at half its bytes on gfx950 (MI355X_MICROARCH.md, HBM)
            e["pass_read_bytes"] = 2.0 * 1024.0 * s["FETCH_SIZE"]["median"]
        if "WRITE_SIZE" in s:
            e["pass_written_bytes"] = 1024.0 * s["WRITE_SIZE"]["median"]
        if "pass_read_bytes" in e and "pass_written_bytes" in e:
            e["pass_hbm_bytes_per_launch"] = e["pass_read_bytes"] + e["pass_written_bytes"]
        e["how"] = ("median over the pass-on-M launches of %s in separate rocprofv3 --pmc runs of bench.py; read = 2 x FETCH_SIZE "
                    "(the gfx950 tally of 128-byte requests at 64 bytes), written = WRITE_SIZE (uncalibrated)" % pk)
        for c in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"):
            if c in s:
                e.setdefault("pass_counters", {})[c] = s[c]["median"]
        mixc = [c for c in s if c.startswith("SQ_INSTS_VALU_")]
        if mixc:   # the pass's VALU instruction mix (wave-instructions per launch, by type)
            e["pass_instruction_mix"] = {c: s[c]["median"] for c in sorted(mixc)}
        if "SQ_LDS_BANK_CONFLICT" in s and "SQ_LDS_IDX_ACTIVE" in s and s["SQ_LDS_IDX_ACTIVE"]["median"] > 0:
            e["pass_lds_conflict_share"] = s["SQ_LDS_BANK_CONFLICT"]["median"] / s["SQ_LDS_IDX_ACTIVE"]["median"]
    rk = next((k for k in summary if k.startswith("k_solve_view_resident")), None)
    if rk:   # the resident solver on a row view: what its waves do (one launch per solve: the median over the run's solves)
        s = summary[rk]
        r = {"kernel": rk}
        for c in ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                  "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE"):
            if c in s:
                r[c] = s[c]["median"]
                r.setdefault("kernel_us_under_the_profiler", s[c]["median_duration_us"])
        if "SQ_WAIT_ANY" in r and r.get("SQ_WAVE_CYCLES", 0) > 0:
            r["waiting_share_of_wave_cycles"] = r["SQ_WAIT_ANY"] / r["SQ_WAVE_CYCLES"]
        if "SQ_LDS_IDX_ACTIVE" in r and "kernel_us_under_the_profiler" in r:
            # SQ_LDS_IDX_ACTIVE sums the LDS-array cycles of every CU that ran a unit; the launch lasts us x clock cycles on
            # each of them (the clock from GRBM_GUI_ACTIVE / wall time where recorded, else 2.4 GHz)
            us = r["kernel_us_under_the_profiler"]
            # (GRBM_GUI_ACTIVE comes back summed over the chip's 8 XCDs)
            clk = (r["GRBM_GUI_ACTIVE"] / 8.0 / (us * 1e-6)) if r.get("GRBM_GUI_ACTIVE", 0) > 0 else 2.4e9
            r["clock_GHz"] = clk / 1e9
            r["lds_busy_share_per_cu_cycle_at_246_units"] = r["SQ_LDS_IDX_ACTIVE"] / (246.0 * us * 1e-6 * clk)
        if "SQ_LDS_BANK_CONFLICT" in r and r.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            r["lds_conflict_share"] = r["SQ_LDS_BANK_CONFLICT"] / r["SQ_LDS_IDX_ACTIVE"]
        e["resident"] = r
    ak = next((k for k in summary if k.startswith("k_affinity_sym")), None) or next((k for k in summary if k.startswith("k_affinity")), None)
    if ak and "SQ_INSTS_VALU" in summary[ak]:
        s = summary[ak]
        insts = s["SQ_INSTS_VALU"]["median"]
        dur_us = s["SQ_INSTS_VALU"]["median_duration_us"]
        simd_cycles = dur_us * 1e-6 * 2.4e9 * 1024.0   # 256 CUs x 4 SIMDs at 2.4 GHz
        e["affinity_issue"] = {
            "kernel": ak, "valu_wave_instructions": insts, "kernel_us_under_the_profiler": dur_us,
            "simd_cycles_available": simd_cycles,
            "frac_at_2_cycles_per_instruction": insts * 2.0 / simd_cycles,   # all of them fp32 (SIMD-32: 2 cycles per wave64)
            "frac_at_4_cycles_per_instruction": insts * 4.0 / simd_cycles,   # all of them fp64 / transcendental-free (16 lanes)
            "note": "SQ_INSTS_VALU x cycles a wave64 VALU instruction holds its SIMD (2: fp32, 4: fp64) / (kernel time x "
                    "1024 SIMDs x 2.4 GHz); the fill mixes an fp32 prefilter with exact fp64 scores, the truth lies between",
        }
        # the instruction mix (separate PMC passes): what pins the fraction between the two bounds above.
        # Cycles a wave64 instruction holds its SIMD-32: 2 for fp32 / int32, 4 for fp64 / int64 / conversions that touch
        # an fp64 operand (counted as CVT: all of them priced at 4), 8 for transcendentals (quarter rate); whatever the
        # typed counters do not cover (moves, compares, permutes, bit ops) at 2.
        types = {"SQ_INSTS_VALU_ADD_F64": 4, "SQ_INSTS_VALU_MUL_F64": 4, "SQ_INSTS_VALU_FMA_F64": 4, "SQ_INSTS_VALU_INT64": 4,
                 "SQ_INSTS_VALU_CVT": 4, "SQ_INSTS_VALU_TRANS_F64": 8, "SQ_INSTS_VALU_TRANS_F32": 8,
                 "SQ_INSTS_VALU_ADD_F32": 2, "SQ_INSTS_VALU_MUL_F32": 2, "SQ_INSTS_VALU_FMA_F32": 2, "SQ_INSTS_VALU_INT32": 2}
        if all(c in s for c in types):
            mix = {c: s[c]["median"] for c in types}
            typed = sum(mix.values())
            cycles = sum(mix[c] * types[c] for c in types) + max(0.0, insts - typed) * 2.0
            e["affinity_issue"]["instruction_mix"] = mix
            e["affinity_issue"]["untyped_instructions"] = insts - typed
            e["affinity_issue"]["frac_by_instruction_mix"] = cycles / simd_cycles
            e["affinity_issue"]["mix_note"] = ("cycles = 4 x (fp64 add/mul/fma, int64, cvt) + 8 x transcendentals + 2 x (fp32 add/mul/fma, "
                                               "int32, and the instructions no typed counter covers), over the SIMD cycles of the launch")
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
            if c in s:
                e["affinity_issue"][c] = s[c]["median"]
    json.dump(rec, open(a.json, "w"), indent=1, sort_keys=True)
    print("wrote", a.json, "key", a.key)


if __name__ == "__main__":
    main()
