#!/usr/bin/env python3
"""Per-launch HBM traffic of the frame kernels from two rocprofv3 --pmc passes.

  pmc_traffic.py <FETCH_SIZE pass dir> <WRITE_SIZE pass dir> <out.json>

FETCH_SIZE and WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE tallies 128-byte read requests
at 64 bytes, so it is multiplied by the factor that tools/calib/fetch_calib.sh measured for the
access shapes of these kernels (profiles/fetch_calib.json: memory is asked for whole 128-byte lines
whether a lane wants 16 bytes of one or all of it, so the factor is 2 for gathers as for streams)
before it is added to WRITE_SIZE.
"""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd._build import source_hash  # the sources these counters were collected on

NAMES = {"k_setup": "setup", "k_raster_shade": "raster_shade", "k_finish": "finish", "k_object": "object",
         "k_clip": "clip", "k_rotation_system": "transform"}

def fetch_factor():
    """The calibrated bytes per counted 64-byte unit / 64 (tools/calib/fetch_calib.sh)."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "fetch_calib.json")
    try:
        f = json.load(open(path)).get("fetch_factor_for_gathers")
    except (OSError, ValueError):
        f = None
    return (float(f), "profiles/fetch_calib.json") if f else (2.0, "uncalibrated: the factor of wide streaming reads")


def main():
    fetch, write = summarise(sys.argv[1]), summarise(sys.argv[2])
    factor, factor_from = fetch_factor()
    out, detail = {}, {}
    for kname, cs in fetch.items():
        key = next((v for k, v in NAMES.items() if k in kname), None)
        if key is None or "FETCH_SIZE" not in cs:
            continue
        w = next((c for n, c in write.items() if n == kname), {})
        f_b = cs["FETCH_SIZE"] * 1024.0
        w_b = w.get("WRITE_SIZE", 0.0) * 1024.0
        detail[key] = {"FETCH_SIZE_bytes_raw": f_b, "FETCH_SIZE_bytes_corrected": factor * f_b, "WRITE_SIZE_bytes": w_b,
                       "launches_averaged": cs["launches"]}
        out[key] = factor * f_b + w_b
    out["detail"] = detail
    out["fetch_factor"] = factor
    out["fetch_factor_from"] = factor_from
    out["note"] = ("bytes per launch = fetch_factor x FETCH_SIZE + WRITE_SIZE (KiB counters, separate --pmc passes, mean over "
                   "the launches of tools/frames.py on config 3); FETCH_SIZE counts 64 bytes per read request of the L2 to "
                   "memory, and a request is a whole 128-byte line for gathered 16-byte reads as for streams "
                   "(tools/calib/fetch_calib.hip: one line read at 16 bytes or at both halves counts the same)")
    out["source_hash"] = source_hash()
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
