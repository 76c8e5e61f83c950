#!/usr/bin/env python3
"""Summarises the passes of tools/pmc_mem.sh: per kernel, the mean of every counter over the
launches of tools/frames.py (config 3), and the busy shares of the vector-memory units.

  pmc_mem.py <directory with the pass directories> <out.json>

A compute unit has one address unit (TA), one data-return unit (TD) and one L1 (TCP); the *_sum
counters add up the 256 of them, in shader-clock cycles.
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd._build import source_hash  # the sources these counters were collected on

CUS = 256

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
                m.setdefault(c, v)
    out = {}
    for k, m in merged.items():
        if not any(s in k for s in ("k_setup", "k_raster_shade", "k_finish", "k_object")):
            continue
        d = dict(m)
        cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # shader-clock cycles of one launch
        if cyc:
            d["launch_cycles"] = cyc
            for c, name in (("TA_TA_BUSY_sum", "ta_busy_frac"), ("TD_TD_BUSY_sum", "td_busy_frac"), ("TCP_GATE_EN1_sum", "tcp_active_frac"),
                            ("TA_ADDR_STALLED_BY_TC_CYCLES_sum", "ta_addr_stalled_by_l1_frac"),
                            ("TA_DATA_STALLED_BY_TC_CYCLES_sum", "ta_data_stalled_by_l1_frac"),
                            ("TCP_PENDING_STALL_CYCLES_sum", "tcp_pending_stall_frac"),
                            ("TCP_TCP_TA_DATA_STALL_CYCLES_sum", "tcp_ta_data_stall_frac")):
                if c in m:
                    d[name] = m[c] / (CUS * cyc)
        if m.get("TA_FLAT_WAVEFRONTS_sum") and "TA_TA_BUSY_sum" in m:
            d["ta_busy_cycles_per_memory_instruction"] = m["TA_TA_BUSY_sum"] / m["TA_FLAT_WAVEFRONTS_sum"]
        out[k] = d
    out["note"] = ("rocprofv3 --pmc, passes of tools/frames.py 3 6 (config 3); every value is the mean over the launches of a "
                   "pass; *_frac = counter / (256 compute units * launch cycles), launch cycles = GRBM_GUI_ACTIVE / 8")
    out["source_hash"] = source_hash()
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    for k, d in out.items():
        if isinstance(d, dict):
            print(k, {c: (round(v, 4) if isinstance(v, float) else v) for c, v in d.items() if not c.startswith("regs")})

if __name__ == "__main__":
    main()
