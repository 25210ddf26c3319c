#!/bin/bash
# usage: scripts/isa_phases.sh [extra flags] : static instruction counts of k_path<PHILOX,false> per phase (between the phase-fence comments)
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DRTW_MIN_WAVES=4 "$@" -S --cuda-device-only -o /tmp/rtw.s $R/raytracing_weekend_amd/csrc/rtw_hip.hip 2>&1 | grep -E "error" 
awk '/^_ZN4rtwk6k_pathILi0ELi0EEEvNS_5KArgsE:/{f=1} f{print} /s_endpgm/{if(f){exit}}' /tmp/rtw.s > /tmp/kpath.s
python3 - <<'PY'
import re,collections
cur="pre"; cnt=collections.OrderedDict()
for l in open('/tmp/kpath.s'):
    l=l.strip()
    m=re.match(r"; MARK (\w+)",l)
    if m: cur=m.group(1); continue
    if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'): continue
    op=l.split()[0]
    c=cnt.setdefault(cur,collections.Counter())
    if op.startswith('v_'): c['valu']+=1
    elif op.startswith('s_load') or op.startswith('s_buffer'): c['smem']+=1
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): c['branch']+=1
    elif op.startswith('s_waitcnt'): c['wait']+=1
    elif op.startswith('s_'): c['salu']+=1
    elif op.startswith('ds_'): c['lds']+=1
    elif op.startswith(('global_','scratch_','buffer_','flat_')): c['vmem']+=1
    if op.startswith('v_mov'): c['v_mov']+=1
    if op.startswith(('v_readlane','v_writelane')): c['lane']+=1
    if op.startswith('v_cndmask'): c['cndmask']+=1
    if op.startswith('v_cmp'): c['cmp']+=1
for k,v in cnt.items(): print(k, dict(v))
PY
