#!/bin/bash
# resource usage table for one instantiation unit: bash tools/resusage.sh attn_bwd_inst.hip 64 [extra flags]
src=$1; d=$2; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -I /root/repo/include -DFAT5_INST_D=$d "$@" -c /root/repo/flasht5_amd/csrc/$src -o /tmp/ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for k in ('VGPRs','AGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]'):
        m=re.search(r'remark:\s+'+k+r': (\d+)',l)
        if m and cur: rows[cur][k[:5]]=m.group(1)
    if 'error' in l: print(l.strip())
for k,v in rows.items():
    name=re.sub(r'_ZN4fat5\d+','',k); name=re.sub(r'EEEvNS_8AttnArgsE','',name)
    print(f'{name:40s} vgpr {v.get(\"VGPRs\")} agpr {v.get(\"AGPRs\")} scratch {v.get(\"Scrat\")} occ {v.get(\"Occup\")}')
"
