#!/bin/bash
# scripts/isa_audit.sh — per kernel of every engine source: registers, scratch, and the four code-generation smells of
# DESIGN §4.1 (scalar spills = v_readlane, generic-address accesses = flat_*, 64-bit VALU address adds, ds_bpermute shuffles)
cd "$(dirname "$0")/../fluid-sims_amd"
echo "# scripts/isa_audit.sh — hipcc $(/opt/rocm/bin/hipcc --version | grep -o 'HIP version: [0-9.-]*'), the Makefile's flags per file; h3d_split = h3d.hip as the"
echo "# split-step translation unit (-DTAU3D_SPLIT_TU, max-ilp scheduler).  waves = floor(512 / VGPRs rounded up to 8), at most 8 per SIMD"
for f in h3d h3d_split h2d sph flow2d stencil2d lbm; do
  src=$f; SPLIT=""
  if [ $f = h3d_split ]; then src=h3d; SPLIT="-DTAU3D_SPLIT_TU -mllvm -amdgpu-sched-strategy=max-ilp"; fi
  EXTRA=$(make -pn 2>/dev/null | grep "^EXTRA_$src " | sed 's/^[^=]*= *//' | sed 's/\$(H3D_DEFS)//')
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize $EXTRA $SPLIT -S --cuda-device-only csrc/$src.hip -o /tmp/audit_$f.s 2>/dev/null
  python3 - /tmp/audit_$f.s $f <<'PY'
import re,sys
txt=open(sys.argv[1]).read(); lines=txt.split('\n')
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel',txt,re.S):
    name=m.group(1); blk=m.group(2)
    g=lambda k:int(re.search(k+r' (\d+)',blk).group(1))
    try:
        start=next(i for i,l in enumerate(lines) if l.startswith(name+':'))
    except StopIteration: continue
    end=next(i for i in range(start,len(lines)) if lines[i].startswith('.Lfunc_end'))
    body=lines[start:end]
    c=lambda s:sum(s in l for l in body)
    import subprocess
    dn=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()[:60]
    v=g('next_free_vgpr'); waves=min(8,512//(((v+7)//8)*8)) if v else 8
    print('%-10s %-60s vgpr %3d sgpr %3d scratch %4d lds %6d waves %d | readlane %3d flat %3d add_u64 %3d bpermute %3d | lines %5d'%(sys.argv[2],dn,v,g('next_free_sgpr'),g('private_segment_fixed_size'),g('group_segment_fixed_size'),waves,c('v_readlane'),c('flat_load')+c('flat_store'),c('v_lshl_add_u64'),c('ds_bpermute'),len(body)))
PY
done
