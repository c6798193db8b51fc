# Round 5: simulated annealing of the LDS slot assignment of the two-worker 16 x 16 Jacobi (csrc/fbx_eigh.hpp, h2_slot): minimises the
# LDS-array cycles of its two ds_write_b128 (served 8 lanes per cycle, bank = 16-byte slot mod 8) and four ds_read_b128 (16 lanes per
# cycle in the groups of MI355X_MICROARCH.md, bank = slot mod 16) per round.  usage: python h2_layout_anneal.py <seed> -> /tmp/layout_<cost>.json
import random, sys
random.seed(int(sys.argv[1]) if len(sys.argv)>1 else 0)
NB=8; PS=66; NSLOT=4*PS
def seat(s):
    k=s>>1
    if s&1==0:
        if k==0: return 0
        if k==NB-1: return 2*(NB-1)+1
        return 2*(k+1)
    if k==0: return 2
    return 2*(k-1)+1
# fixed: diagonal blocks at the standard layout (planes 0,1,3 at K*9); zero cell at slot 56
fixed={}
for K in range(8):
    for e in (0,1,3): fixed[(K,K,e)]=e*PS+K*9
ZERO=56
items=[(I,J,e) for I in range(8) for J in range(I+1,8) for e in range(4)]
used=set(fixed.values())|{ZERO}
free=[s for s in range(NSLOT) if s not in used]
# lane tables
W=[[None,None] for _ in range(64)]; R=[[None]*4 for _ in range(64)]
zs={}
for lane in range(64):
    Ir,Jc=lane//8,lane%8
    diag=Ir==Jc; wb=Ir>Jc
    I,J=(Jc,Ir) if wb else (Ir,Jc); col=1 if wb else 0
    for a in range(2):
        if diag: r2=seat(2*Ir+a); c2=r2
        else: r2=seat(2*I+a); c2=seat(2*J+col)
        I2,J2,a2,b2=r2>>1,c2>>1,r2&1,c2&1
        fl = I2>J2 or (I2==J2 and a2>b2)
        W[lane][a]=(J2,I2,b2*2+a2) if fl else (I2,J2,a2*2+b2)
    for e in range(4):
        a=e>>1; b=(e&1)^col
        key=(I,J,a*2+b)
        zero=False
        if not diag:
            if I==0 and J==1: zero = a==0 and b==0
            elif J==I+2: zero = a==1 and b==0
            elif I==6 and J==7: zero = a==1 and b==1
        R[lane][e]=None if zero else key
wg=[list(range(8*g,8*g+8)) for g in range(8)]
g0=[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27]; g1=[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]
rg=[g0,g1,[x+32 for x in g0],[x+32 for x in g1]]
def slot(assign,key):
    if key is None: return ZERO
    if key in fixed: return fixed[key]
    if key[0]==key[1]: return 2*PS+key[0]*9   # plane 2 of a diagonal block: never used meaningfully
    return assign[key]
def cost(assign):
    c=0
    for a in range(2):
        for g in wg:
            banks={}
            for l in g:
                s=slot(assign,W[l][a]); banks.setdefault(s%8,set()).add(s)
            c+=max(len(v) for v in banks.values())
    for e in range(4):
        for g in rg:
            banks={}
            for l in g:
                s=slot(assign,R[l][e]); banks.setdefault(s%16,set()).add(s)
            c+=max(len(v) for v in banks.values())
    return c
# start: standard layout
assign={(I,J,e):e*PS+I*8+J for (I,J,e) in items}
occupied={v:k for k,v in assign.items()}
cur=cost(assign); best=cur; bestA=dict(assign)
import math
T=2.0
for it in range(1200000):
    k=random.choice(items)
    s_new=random.choice(free)
    s_old=assign[k]
    if s_new==s_old: continue
    other=occupied.get(s_new)
    assign[k]=s_new
    if other is not None: assign[other]=s_old
    c=cost(assign)
    if c<=cur or random.random()<math.exp((cur-c)/T):
        cur=c
        occupied[s_new]=k
        if other is not None: occupied[s_old]=other
        else: occupied.pop(s_old,None)
        if c<best: best=c; bestA=dict(assign)
    else:
        assign[k]=s_old
        if other is not None: assign[other]=s_new
    T=max(0.05,T*0.999996)
    if best<=32: break
print("best cost",best,"(ideal 32; standard layout", cost({(I,J,e):e*PS+I*8+J for (I,J,e) in items}),")")
import json
json.dump({f"{k[0]},{k[1]},{k[2]}":v for k,v in bestA.items()}, open(f"/tmp/layout_{best}.json","w"))
