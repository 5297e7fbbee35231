#!/bin/bash
# ISA summary of one h3d kernel's march loop: usage scripts/k2_isa.sh [mangled-name-prefix]
cd /root/repo/fluid-sims_amd && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize $H3D_DEFS -S --cuda-device-only csrc/h3d.hip -o /tmp/h3d.s 2>&1 | grep -E "error" ; cd /root/repo
python - "$@" <<'PY'
import re,sys
sys.path.insert(0,'scripts')
import isa_mix
from collections import Counter
name=sys.argv[1] if len(sys.argv)>1 else '_ZN3h3d10k_update_zILb1ELb0EEEvNS_4ArgsE'
txt=open('/tmp/h3d.s').read()
lines=txt.split('\n')
start=next(i for i,l in enumerate(lines) if l.startswith(name+':'))
end=next(i for i in range(start,len(lines)) if lines[i].startswith('.Lfunc_end'))
body=lines[start:end]
print('readlane',sum('v_readlane' in l for l in body),'writelane',sum('v_writelane' in l for l in body))
for h in [i for i,l in enumerate(body) if 'Inner Loop Header' in l]:
    lab=body[h].split(':')[0]
    bk=[i for i,l in enumerate(body) if re.search(r'branch\S*\s+%s\b'%re.escape(lab),l)]
    if not bk: continue
    seg=body[h:max(bk)+1]
    if len(seg)<100: continue
    c=Counter(); ops=Counter()
    for l in seg:
        k=isa_mix.classify(l)
        if k:
            c[k]+=1
            if k in('lane',) or k=='half:sgpr': ops[l.split()[0]]+=1
    valu=sum(v for k,v in c.items() if k in('full','trans','lane') or k.startswith('half'))
    half=sum(v for k,v in c.items() if k.startswith('half'))
    est=2.3*c['full']+4.4*half+10*c['trans']+4.4*c['lane']
    print('lines',len(seg),'VALU',valu,'full',c['full'],'half',half,'trans',c['trans'],'lane',c['lane'],'est cycles %.0f'%est, dict(c))
    print('   ',ops.most_common(8))
m=re.search(r'\.amdhsa_kernel '+name+r'.*?\.end_amdhsa_kernel',txt,re.S)
for k in ('next_free_vgpr','next_free_sgpr','private_segment_fixed_size'):
    print(k,re.search(k+r' (\d+)',m.group(0)).group(1))
PY
