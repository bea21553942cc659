# LDS bank-conflict simulation for the unpadded DMA layout: piece (R, c) at byte R*CH*16 + (c ^ x(R))*16
import itertools
B128_GROUPS=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
B128_GROUPS+= [[l+32 for l in g] for g in B128_GROUPS]
def k_conflicts(CH, x):
    worst=1
    KS=CH//4
    for t in range(4):
        for ks in range(KS):
            for g in B128_GROUPS:
                banks={}
                for lane in g:
                    fr,fq=lane&15,lane>>4
                    R=16*t+fr; c=4*ks+fq
                    a=R*CH*16+((c^x(R)))*16
                    for d in range(4):
                        b=(a//4+d)%64
                        banks.setdefault(b,set()).add(a)
                worst=max(worst,max(len(v) for v in banks.values()))
    return worst
def v_conflicts(CH, x):
    worst=1
    DT=CH//2
    for s2e in range(4):
        for dt in range(DT):
            for g in ([*range(32)],[*range(32,64)]):
                banks={}
                for lane in g:
                    fr,fq=lane&15,lane>>4
                    tq,tp=fr>>2,fr&3
                    R=16*s2e+4*fq+tq
                    c=2*dt+(tp>>1)
                    a=R*CH*16+((c^x(R)))*16+8*(tp&1)
                    for d in range(2):
                        b=(a//4+d)%64
                        banks.setdefault(b,set()).add(a)
                worst=max(worst,max(len(v) for v in banks.values()))
    return worst
for CH in (4,8,12,16):
    lim = 4 if CH in (4,12) else (8 if CH==8 else 16)   # xor range keeping c^x inside the row (12: blocks of 4)
    best=None
    # candidate family: x(R) = (a*(R>>s)) & (lim-1) for small a, s, plus xor of two such
    cands=[]
    for s1 in range(0,5):
        for a1 in range(0,lim):
            for s2 in range(0,5):
                for a2 in range(0,lim):
                    cands.append((s1,a1,s2,a2))
    resK=[];resV=[]
    for (s1,a1,s2,a2) in cands:
        x=lambda R,s1=s1,a1=a1,s2=s2,a2=a2: (((R>>s1)*a1) ^ ((R>>s2)*a2)) & (lim-1)
        if not resK or True:
            k=k_conflicts(CH,x)
            if k==1: resK.append((s1,a1,s2,a2))
        v=v_conflicts(CH,x)
        if v==1: resV.append((s1,a1,s2,a2))
        if k==1 and v==1 and best is None: best=(s1,a1,s2,a2)
    print("CH",CH,"identity K",k_conflicts(CH,lambda R:0),"V",v_conflicts(CH,lambda R:0),"| K ok:",resK[:4],"| V ok:",resV[:4],"| both:",best)
