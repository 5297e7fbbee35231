#!/bin/bash
# scripts/isa_h3d.sh [extra -D flags] — registers / scratch / LDS / instruction lines of the 3D step's kernels only (isa_audit.sh, h3d rows)
cd "$(dirname "$0")/../fluid-sims_amd"
for f in h3d h3d_split; do
  SPLIT=""; [ $f = h3d_split ] && SPLIT="-DTAU3D_SPLIT_TU -mllvm -amdgpu-sched-strategy=max-ilp"
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -ffp-contract=on "$@" $SPLIT -S --cuda-device-only csrc/h3d.hip -o /tmp/isa_$f.s 2>/dev/null &
done
wait
for f in h3d h3d_split; do
python3 - /tmp/isa_$f.s $f <<'PY'
import re,sys,subprocess
txt=open(sys.argv[1]).read(); lines=txt.split('\n')
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel',txt,re.S):
    name=m.group(1); blk=m.group(2)
    dn=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()[:48]
    if not any(k in dn for k in ('k_step','k_flux_xy','k_update_z')): continue
    g=lambda k:int(re.search(k+r' (\d+)',blk).group(1))
    start=next(i for i,l in enumerate(lines) if l.startswith(name+':'))
    end=next(i for i in range(start,len(lines)) if lines[i].startswith('.Lfunc_end'))
    body=[l.strip() for l in lines[start:end]]
    valu=[l for l in body if l.startswith('v_')]
    half=sum(1 for l in valu if re.match(r'v_(max|min|med3|cndmask|cmp|bfi|and_or|mov_b32_dpp|ldexp|frexp)',l) or '_dpp' in l)
    trans=sum(1 for l in valu if re.match(r'v_(rcp|rsq|sqrt|exp|log)_',l))
    v=g('next_free_vgpr')
    print('%-10s %-48s vgpr %3d scratch %4d lds %6d | valu %5d (half %4d trans %3d) cndmask %4d ds %4d'%(sys.argv[2],dn,v,g('private_segment_fixed_size'),g('group_segment_fixed_size'),len(valu),half,trans,sum(l.startswith('v_cndmask') for l in valu),sum(l.startswith('ds_') for l in body)))
PY
done
