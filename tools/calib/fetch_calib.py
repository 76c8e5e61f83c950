#!/usr/bin/env python3
"""Prices FETCH_SIZE against the known line counts of tools/calib/fetch_calib.hip.

  fetch_calib.py <directory of tools/calib/fetch_calib.sh> <out.json>

Per case: the counter in bytes (KiB x 1024), and what one counted unit of 64 bytes stood for if
memory was asked for whole 128-byte lines, or for 64-byte half lines.  The rule that fits every
case is the one to correct the tile kernel's FETCH_SIZE with."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pmc_summary import summarise


def main():
    root, out_path = sys.argv[1], sys.argv[2]
    plain = json.load(open(os.path.join(root, "plain.json")))
    fetch = summarise(os.path.join(root, "fetch"))
    rdreq = summarise(os.path.join(root, "rdreq")) if os.path.isdir(os.path.join(root, "rdreq")) else {}
    out = {"buffer_bytes": plain["buffer_bytes"], "cases": {}}
    for name, c in plain["cases"].items():
        f = next((v for k, v in fetch.items() if k.startswith(name)), None)
        if f is None:
            continue
        counted = f["FETCH_SIZE"] * 1024.0
        d = dict(c)
        d["FETCH_SIZE_bytes"] = counted
        d["requests_at_64B"] = counted / 64.0
        d["requests_per_line"] = counted / 64.0 / c["lines_128B"]
        d["requests_per_half_line"] = counted / 64.0 / c["half_lines_64B"]
        d["GBps_if_requests_are_128B"] = counted * 2.0 / (c["ms"] * 1e-3) / 1e9
        d["GBps_if_requests_are_64B"] = counted / (c["ms"] * 1e-3) / 1e9
        r = next((v for k, v in rdreq.items() if k.startswith(name)), None)
        if r:
            d["TCC_EA0_RDREQ_sum"] = r.get("TCC_EA0_RDREQ_sum")
            d["TCC_EA0_RDREQ_32B_sum"] = r.get("TCC_EA0_RDREQ_32B_sum")
        out["cases"][name] = d
    one, both = out["cases"].get("gather16_one_per_line"), out["cases"].get("gather16_both_halves")
    if one and both:
        ratio = both["FETCH_SIZE_bytes"] / one["FETCH_SIZE_bytes"]
        out["both_halves_over_one_half"] = ratio
        # the second half of a line costs nothing more: memory is asked for whole lines, 128 bytes
        # per counted request, and the counter is doubled; it costs as much again: half lines, 64
        # bytes per counted request, and the counter stands as it is
        out["fetch_factor_for_gathers"] = 2.0 if ratio < 1.25 else (1.0 if ratio > 1.75 else None)
        out["fetch_factor_for_streams"] = out["cases"]["stream16"]["bytes_asked"] / out["cases"]["stream16"]["FETCH_SIZE_bytes"]
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
