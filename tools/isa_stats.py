"""Static instruction mix of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).
usage: python tools/isa_stats.py listing.s kernel_symbol"""
import sys
from collections import Counter

def stats(path, name):
    s = open(path).read()
    i = s.index("\n" + name + ":")
    j = s.index("\n.Lfunc_end", i)
    ins = []
    for l in s[i:j].split("\n"):
        w = l.split()
        if l.startswith("\t") and w and not w[0].startswith((".", ";")):
            ins.append(w[0])
    c = Counter()
    for x in ins:
        c["valu" if x.startswith("v_") else "salu" if x.startswith("s_") else "lds" if x.startswith("ds_") else "mem"] += 1
    c["v_mov"] = sum(1 for x in ins if x.startswith("v_mov"))
    c["v_cndmask"] = sum(1 for x in ins if x.startswith("v_cndmask"))
    c["total"] = len(ins)
    return dict(c)

if __name__ == "__main__":
    print(stats(sys.argv[1], sys.argv[2]))
