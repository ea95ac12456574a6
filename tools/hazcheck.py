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
print("hazards:",bad)
