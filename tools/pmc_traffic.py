#!/usr/bin/env python3
"""Per-launch HBM traffic of the frame kernels from two rocprofv3 --pmc passes.

  pmc_traffic.py <FETCH_SIZE pass dir> <WRITE_SIZE pass dir> <out.json>

FETCH_SIZE and WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE tallies 128-byte read requests
at 64 bytes, so it is doubled before it is added to WRITE_SIZE.
"""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd._build import source_hash  # the sources these counters were collected on

NAMES = {"k_setup": "setup", "k_raster_shade": "raster_shade", "k_finish": "finish", "k_object": "object",
         "k_clip": "clip", "k_rotation_system": "transform"}

def main():
    fetch, write = summarise(sys.argv[1]), summarise(sys.argv[2])
    out, detail = {}, {}
    for kname, cs in fetch.items():
        key = next((v for k, v in NAMES.items() if k in kname), None)
        if key is None or "FETCH_SIZE" not in cs:
            continue
        w = next((c for n, c in write.items() if n == kname), {})
        f_b = cs["FETCH_SIZE"] * 1024.0
        w_b = w.get("WRITE_SIZE", 0.0) * 1024.0
        detail[key] = {"FETCH_SIZE_bytes_raw": f_b, "FETCH_SIZE_bytes_corrected": 2.0 * f_b, "WRITE_SIZE_bytes": w_b,
                       "launches_averaged": cs["launches"]}
        out[key] = 2.0 * f_b + w_b
    out["detail"] = detail
    out["note"] = ("bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters, separate --pmc passes, mean over the "
                   "launches of tools/frames.py on config 3); the doubling is calibrated for wide streaming reads and is "
                   "an upper bound for the gathered 16-byte reads of these kernels")
    out["source_hash"] = source_hash()
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
