#!/usr/bin/env python3
"""Summarises the passes of tools/pmc_valu_lds.sh: per kernel, the mean of every counter over the
launches of tools/frames.py (config 3), plus the figures derived from them.

  pmc_valu_lds.py <directory with the pass directories> <out.json>

Units (gfx950): SQ_*_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count in units of four cycles, summed
over all SIMDs (wave-level counters) of the chip; GRBM_GUI_ACTIVE is summed over the eight XCDs.
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd._build import source_hash  # the sources these counters were collected on

SIMDS = 256 * 4

def main():
    root, out_path = sys.argv[1], sys.argv[2]
    merged = {}
    for name in sorted(os.listdir(root)):
        d = os.path.join(root, name)
        if not os.path.isdir(d):
            continue
        for k, cs in summarise(d).items():
            short = k.split("<")[0].replace("void ", "").replace("sxe::", "")
            m = merged.setdefault(short, {})
            for c, v in cs.items():
                if c == "GRBM_GUI_ACTIVE" and c in m:
                    m[c] = 0.5 * (m[c] + v) if isinstance(v, float) else v
                else:
                    m.setdefault(c, v)
    out = {}
    for k, m in merged.items():
        if not any(s in k for s in ("k_setup", "k_raster_shade", "k_finish", "k_object", "k_clip")):
            continue
        d = dict(m)
        cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # shader-clock cycles of one launch
        if cyc and "SQ_ACTIVE_INST_VALU" in m:
            # time the vector ALUs were issuing, as a fraction of what all SIMDs offer over the launch
            d["valu_busy_frac"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (SIMDS * cyc)
            d["cycles_per_valu_inst"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / max(m.get("SQ_INSTS_VALU", 0.0), 1.0)
        if "SQ_THREAD_CYCLES_VALU" in m and m.get("SQ_ACTIVE_INST_VALU"):
            d["valu_lane_use"] = m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"])
        if cyc and "SQ_WAVE_CYCLES" in m:
            d["mean_waves_per_simd"] = 4.0 * m["SQ_WAVE_CYCLES"] / (SIMDS * cyc)
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
                if c in m:
                    d[c.lower() + "_share_of_wave_cycles"] = m[c] / m["SQ_WAVE_CYCLES"]
        if cyc and "SQ_LDS_IDX_ACTIVE" in m:
            d["lds_busy_frac"] = m["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)  # LDS-array cycles of 256 CUs
            d["lds_bank_conflict_share"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(m["SQ_LDS_IDX_ACTIVE"], 1.0)
        if cyc:
            d["launch_cycles"] = cyc
        out[k] = d
    out["note"] = ("rocprofv3 --pmc, three passes of tools/frames.py 3 6 (config 3); every value is the mean over the "
                   "launches of a pass; valu_busy_frac = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * launch cycles), launch "
                   "cycles = GRBM_GUI_ACTIVE / 8")
    out["source_hash"] = source_hash()
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    for k, d in out.items():
        if isinstance(d, dict):
            print(k, {c: (round(v, 4) if isinstance(v, float) else v) for c, v in d.items() if not c.startswith("regs")})

if __name__ == "__main__":
    main()
