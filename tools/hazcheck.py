"""developer aid: scan hipcc -S output for hazards the compiler cannot see through inline asm (VALU-written SGPR read by a
VMEM instruction within 5 wait states; vector registers handed to scalar instructions).  usage: hazcheck.py file.s"""
import re,sys
s=open(sys.argv[1]).read().split('\n')
# flatten instruction stream with indices
ins=[(i,l.strip()) for i,l in enumerate(s) if l.strip() and not l.strip().startswith((';','.','//')) and not l.strip().endswith(':')]
bad=0
for n,(i,t) in enumerate(ins):
    m=re.match(r'(global_load_lds_dwordx4|global_store_dwordx4|global_load_dwordx4)\s+.*s\[(\d+):(\d+)\]',t)
    if not m: continue
    a,b=int(m.group(2)),int(m.group(3))
    # count wait states back
    ws=0
    for k in range(n-1,max(n-8,-1),-1):
        tt=ins[k][1]
        mm=re.match(r'(v_readlane_b32|v_readfirstlane_b32)\s+s(\d+)',tt)
        if mm and int(mm.group(2)) in (a,b) and ws<5:
            bad+=1; print("HAZARD line",i,t,"<-",tt,"ws",ws); break
        mn=re.match(r's_nop\s+(\d+)',tt)
        ws+= (int(mn.group(1))+1) if mn else 1
# gfx940+: a VGPR written by a VALU instruction needs 1 wait state before v_readlane / v_readfirstlane reads it
for n,(i,t) in enumerate(ins):
    m=re.match(r'(v_readlane_b32|v_readfirstlane_b32)\s+s\d+,\s*v(\d+)',t)
    if not m or n==0: continue
    pt=ins[n-1][1]
    mm=re.match(r'v_\w+\s+v(\d+)\b',pt)
    if mm and mm.group(1)==m.group(2) and not pt.startswith(('v_readlane','v_readfirstlane')):
        bad+=1; print("HAZARD line",i,t,"<- VGPR written by the previous instruction:",pt)
for i,t in ins:
    if re.match(r's_(mov|and|or|andn2)_b(32|64)\s+(exec|m0|s\[?\d+).*\bv\[?\d+', t):
        bad+=1; print("VGPR operand on a scalar instruction, line",i,t)
print("hazards:",bad)
