"""Per-loop register-pressure symptoms of a hipcc -S listing: for every loop that contains MFMAs, the count of scratch accesses, `s_waitcnt vmcnt(0)` and
vector moves inside it.  A scratch reload inside a K loop that keeps LDS-DMA in flight is a drained ring (the reload is a vector-memory operation: hipcc waits
vmcnt(0) for it): the 8-second loop used to get gemm256's MIXC instances clean.
  hipcc --offload-arch=gfx950 -O3 -std=c++20 -Iarp_amd/csrc -S --cuda-device-only scripts/mixc_isa.hip -o /tmp/mixc.s && python scripts/isa_loops.py /tmp/mixc.s"""
import re,sys
lines=open(sys.argv[1]).read().split('\n')
# find loops: label lines followed by "Inner Loop Header", back-edge = s_cbranch to that label
labels={}
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: labels[m.group(1)]=i
for i,l in enumerate(lines):
    if 'Inner Loop Header' in l:
        # label is previous line(s)
        j=i
        while j>=0 and not re.match(r'^(\.LBB\d+_\d+):',lines[j]): j-=1
        lab=re.match(r'^(\.LBB\d+_\d+):',lines[j]).group(1)
        # last branch to lab
        ends=[k for k in range(j,len(lines)) if re.search(r's_cbranch\w* '+re.escape(lab)+r'\b',lines[k]) or re.search(r's_branch '+re.escape(lab)+r'\b',lines[k])]
        if not ends: continue
        e=max(ends)
        body=lines[j:e+1]
        print(lab, 'lines',j,e,'mfma',sum('v_mfma' in x for x in body),'scratch',sum('scratch_' in x for x in body),'vmcnt0',sum('vmcnt(0)' in x for x in body), 'v_mov', sum('v_mov_b' in x for x in body))
