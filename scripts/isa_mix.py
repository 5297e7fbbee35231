#!/usr/bin/env python3
"""isa_mix.py <file.s> <kernel-symbol-substring> — instruction mix of every innermost loop of one kernel,
classified by the issue cost measured with scripts/valu_calib.hip on an MI355X (profiles/r02/valu_calib.txt):

  full   ~2.2-2.5 cycles / wave64 instr / SIMD : fma, fmac, mul, add, sub, mov, integer add / and / shifts with VGPR, inline-constant
                                                 or 32-bit-literal sources
  half   ~4.3-4.7 cycles                        : the same opcodes with an SGPR source; v_max / v_min / v_med3; v_cmp*; v_cndmask;
                                                 any DPP form
  trans  ~8.2 cycles (12 inside an FMA stream)   : v_rcp, v_rsq, v_sqrt, v_exp, v_log
Static counts of the loop body (conditional blocks included at weight 1).
"""
import re
import sys
from collections import Counter

TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
HALF_OPS = ("v_max_", "v_min_", "v_med3_", "v_cmp", "v_cndmask", "v_max3", "v_min3")


def classify(line):
    t = line.split(";")[0].split()
    if not t:
        return None
    op = t[0]
    if op.startswith(("s_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
        return op.split("_")[0] if not op.startswith("s_") else ("s_waitcnt" if op == "s_waitcnt" else "salu")
    if not op.startswith("v_"):
        return None
    if op.startswith(TRANS):
        return "trans"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    args = " ".join(t[1:])
    if "dpp" in op or "row_" in args or "wave_sh" in args or "quad_perm" in args:
        return "half:dpp"
    if op.startswith(HALF_OPS):
        return "half:" + op.split("_")[1]
    srcs = args.split(",")[1:]
    if any(re.match(r"\s*-?\|?(s\d+|s\[\d+:\d+\]|vcc|exec|m0)", s) for s in srcs) and not op.startswith("v_cndmask"):
        return "half:sgpr"
    return "full"


def main():
    path, sym = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^\S*%s\S*:" % re.escape(sym), l))
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith("\t.section"))
    body = lines[start:end]
    heads = [i for i, l in enumerate(body) if "Loop Header" in l]
    for h in heads:
        label = body[h].split(":")[0]
        back = max(i for i, l in enumerate(body) if re.search(r"s_c?branch\S*\s+%s\b" % re.escape(label), l))
        c = Counter()
        ops = Counter()
        for l in body[h:back + 1]:
            k = classify(l)
            if k:
                c[k] += 1
                if k.startswith("half:sgpr"):
                    ops[l.split()[0]] += 1
        valu = sum(v for k, v in c.items() if k in ("full", "trans", "lane") or k.startswith("half"))
        half = sum(v for k, v in c.items() if k.startswith("half"))
        est = 2.3 * c["full"] + 4.4 * half + 10.0 * c["trans"] + 4.4 * c["lane"]
        print(f"loop {label} (lines {h}-{back}): VALU {valu}  full {c['full']}  half {half}  trans {c['trans']}  | est {est:.0f} cycles, {est / max(valu, 1):.2f} per instr")
        print("   ", ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
        print("    sgpr-source opcodes:", ", ".join(f"{k} {v}" for k, v in ops.most_common(8)))


if __name__ == "__main__":
    main()
