"""Audit of the -save-temps ISA of conv_igemm.hip: no instruction may touch the destination registers of a hidden
(inline-asm) global load between the load and the next counted s_waitcnt vmcnt.  Usage: hipcc ... -save-temps=obj, then run."""
import re
s=open('/tmp/conv_igemm-hip-amdgcn-amd-amdhsa-gfx950.s').read()
bad=0
for m in re.finditer(r'^(_ZN2pp17conv_split_kernel\w+):', s, flags=re.M):
    name=m.group(1)
    a=m.start(); b=s.index('.Lfunc_end',a)
    body=[l.strip() for l in s[a:b].split('\n')]
    pending={}
    for i,l in enumerate(body):
        if l.startswith('global_load_dwordx4') and body[i-1].startswith(';;#ASMSTART'):
            mm=re.match(r'global_load_dwordx4 v\[(\d+):(\d+)\]',l)
            for r in range(int(mm.group(1)),int(mm.group(2))+1): pending[r]=i
            continue
        if l.startswith('s_waitcnt') and 'vmcnt' in l:
            pending.clear(); continue
        if not l or l[0] in '.;': continue
        regs=set()
        for mm in re.finditer(r'v\[(\d+):(\d+)\]',l): regs.update(range(int(mm.group(1)),int(mm.group(2))+1))
        for mm in re.finditer(r'\bv(\d+)\b',l): regs.add(int(mm.group(1)))
        hit=regs & set(pending)
        if hit:
            bad+=1
            if bad<12: print(name[-36:], i, l, sorted(hit)[:4])
print('suspicious', bad)
