#!/usr/bin/env python3
"""isa_kernel_mix.py <file.s> <kernel-symbol-substring> — static instruction mix of a whole kernel (every block at weight 1),
classes as in isa_mix.py.  For straight-line kernels (k_flux_xy); compile with -DTAU3D_FAST_ONLY so that only the body that
runs is in the object."""
import re
import sys
from collections import Counter
sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_mix import classify

path, sym = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^\S*%s\S*:" % re.escape(sym), l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
c = Counter()
ops = Counter()
for l in lines[start:end]:
    k = classify(l)
    if k:
        c[k] += 1
        ops[(k, l.split()[0])] += 1
valu = sum(v for k, v in c.items() if k in ("full", "trans", "lane") or k.startswith("half"))
half = sum(v for k, v in c.items() if k.startswith("half"))
cyc = sum(v * (2.3 if k == "full" else 4.5 if k.startswith("half") or k == "lane" else 10.0 if k == "trans" else 0) for k, v in c.items())
print(f"{sym}: VALU {valu} (full {c['full']}, half {half}, trans {c['trans']}, lane {c['lane']})  est. issue cycles {cyc:.0f} = {cyc/max(valu,1):.2f}/instr"
      f" | ds {c['ds']} global {c['global']} salu {c['salu']} waitcnt {c['s_waitcnt']}")
print("  half:", dict((k, v) for k, v in c.items() if k.startswith("half")))
if len(sys.argv) > 3:
    for (k, op), v in sorted(ops.items(), key=lambda kv: -kv[1])[:40]:
        print(f"   {v:5d} {k:10s} {op}")
