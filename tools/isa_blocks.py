#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel's ISA (VALU / transcendental / MFMA / LDS / SALU / VMEM): what a wave issues
where, when no PC sampling is to be had.  Input: the kernel's slice of hipcc's -save-temps .s file, e.g.

    hipcc -O3 --offload-arch=gfx950 -Iinclude -Iremora_amd/csrc -c remora_amd/csrc/k_fused.hip -save-temps -o /tmp/x.o
    awk '/^_ZN3rmr18fused_front_kernelILi9ELb0EEEvNS_9FusedArgsE:/,/s_endpgm/' k_fused-hip-amdgcn-amd-amdhsa-gfx950.s > k9.s
    python tools/isa_blocks.py k9.s"""
import re,sys
blk=None; order=[]; cnt={}
for ln in open(sys.argv[1]):
    m=re.match(r'^(\.LBB\d+_\d+):',ln)
    if m or 's_barrier' in ln:
        blk = m.group(1) if m else (blk+"+bar")
        if blk not in cnt: cnt[blk]=dict(valu=0,mfma=0,lds=0,salu=0,trans=0,vmem=0); order.append(blk)
        continue
    if blk is None:
        blk='entry'; cnt[blk]=dict(valu=0,mfma=0,lds=0,salu=0,trans=0,vmem=0); order.append(blk)
    t=ln.strip().split()
    if not t or t[0].startswith(';') or t[0].startswith('.'): continue
    op=t[0]
    c=cnt[blk]
    if op.startswith('v_mfma'): c['mfma']+=1
    elif op.startswith('v_exp') or op.startswith('v_rcp'): c['trans']+=1; c['valu']+=1
    elif op.startswith('v_'): c['valu']+=1
    elif op.startswith('ds_'): c['lds']+=1
    elif op.startswith('s_'): c['salu']+=1
    elif op.startswith('buffer') or op.startswith('global'): c['vmem']+=1
for b in order:
    c=cnt[b]
    if sum(c.values())>8: print(f"{b:16s} valu {c['valu']:4d} (trans {c['trans']:3d}) mfma {c['mfma']:3d} lds {c['lds']:3d} salu {c['salu']:3d} vmem {c['vmem']:3d}")
