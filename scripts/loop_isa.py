#!/usr/bin/env python3
"""ISA mix of the loops of one kernel in a hipcc -S listing: scripts/loop_isa.py file.s mangled_kernel_name
(classes and cycle weights: scripts/isa_mix.py, calibrated in profiles/r02/valu_calib.txt)"""
import re, sys
from collections import Counter
sys.path.insert(0, __file__.rsplit('/', 1)[0])
import isa_mix

def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ':'))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = lines[start:end]
    print('kernel lines', len(body), 'readlane', sum('v_readlane' in l for l in body), 'scratch', sum('scratch_' in l for l in body),
          'flat', sum('flat_' in l for l in body), 'lshl_add_u64', sum('v_lshl_add_u64' in l for l in body))
    labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
    for lab, h in labels.items():
        bk = [i for i, l in enumerate(body) if i > h and re.search(r's_c?branch\S*\s+%s\b' % re.escape(lab), l)]
        if not bk: continue
        seg = body[h:max(bk) + 1]
        if len(seg) < 100: continue
        c = Counter(); ops = Counter()
        for l in seg:
            k = isa_mix.classify(l)
            if k:
                c[k] += 1
                if k in ('lane', 'half:sgpr'): ops[l.split()[0]] += 1
        valu = sum(v for k, v in c.items() if k in ('full', 'trans', 'lane') or k.startswith('half'))
        half = sum(v for k, v in c.items() if k.startswith('half'))
        est = 2.3 * c['full'] + 4.4 * half + 10 * c['trans'] + 4.4 * c['lane']
        print(lab, 'lines', len(seg), 'VALU', valu, 'full', c['full'], 'half', half, 'trans', c['trans'], 'lane', c['lane'], 'est cycles %.0f' % est)
        print('   ', dict(c)); print('   ', ops.most_common(10))

main()
