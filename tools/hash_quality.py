"""Statistical quality of dropout-hash candidates (CPU, numpy): avalanche over every input bit on random inputs, and keep rate / neighbour,
row and key correlations of the 16-bit halves on sequential indices.  `candC` is what mmf_amd/csrc/common.h::mix24 implements; `current`
is the 32-bit-multiply finalizer it replaced in round 3.    python tools/hash_quality.py"""
import numpy as np
M32=np.uint64(0xFFFFFFFF)
def u(x): return np.asarray(x,dtype=np.uint64)&M32
def mul24(a,b): return ((u(a)&np.uint64(0xFFFFFF))*(u(b)&np.uint64(0xFFFFFF)))&M32
def mix32(x):
    x=u(x); x^=x>>np.uint64(16); x=(x*np.uint64(0x7feb352d))&M32; x^=x>>np.uint64(15); x=(x*np.uint64(0x846ca68b))&M32; x^=x>>np.uint64(16); return x
def cur(key,idx): return mix32((u(idx)*np.uint64(0x9E3779B1)+np.uint64(key))&M32)
def rotl(x,r): x=u(x); return ((x<<np.uint64(r))|(x>>np.uint64(32-r)))&M32
# candidate: 3 u24 multiplies + xorshifts (all full-rate ops)
def candA(key,idx):
    x=(u(idx)+np.uint64(key))&M32
    x^=x>>np.uint64(15)
    x=(mul24(x,0xB5297B)+ (x>>np.uint64(24)))&M32 ^ rotl(x,11)
    x^=x>>np.uint64(13)
    x=(mul24(x,0x68E31F)+(x>>np.uint64(24)))&M32 ^ rotl(x,7)
    x^=x>>np.uint64(16)
    x=(mul24(x,0x1B873B)^ (x>>np.uint64(9)))&M32
    x^=x>>np.uint64(14)
    return x
def candB(key,idx):   # Feistel on 16-bit halves, 4 rounds, F = mul24 high bits
    x=(u(idx)+np.uint64(key))&M32
    L=x>>np.uint64(16); R=x&np.uint64(0xFFFF)
    for C in (0xB5297B,0x68E31F,0x1B873B,0x9E3779):
        F=(mul24(R+np.uint64(0x5A17),C)>>np.uint64(8))&np.uint64(0xFFFF)
        L,R=R,(L^F)
    return ((L<<np.uint64(16))|R)&M32
def avalanche(h,n=20000,seed=0):
    rng=np.random.default_rng(seed)
    idx=rng.integers(0,2**31,n,dtype=np.uint64); key=int(rng.integers(0,2**32))
    base=h(key,idx)
    worst=0;mat=[]
    for b in range(31):
        d=base^h(key,idx^np.uint64(1<<b))
        p=np.array([((d>>np.uint64(o))&np.uint64(1)).mean() for o in range(32)])
        mat.append(p)
    mat=np.array(mat); return float(np.abs(mat-0.5).max()), float(np.abs(mat-0.5).mean())
def seq_stats(h,thr=6554):
    key=0x1234567
    idx=np.arange(0,1<<20,dtype=np.uint64)
    hh=h(key,idx)
    lo=(hh&np.uint64(0xFFFF)); hi=hh>>np.uint64(16)
    keep=np.stack([lo>=thr,hi>=thr],1).reshape(-1).astype(np.float64)
    rate=keep.mean()
    c1=np.corrcoef(keep[:-1],keep[1:])[0,1]; c2=np.corrcoef(keep[:-256],keep[256:])[0,1]; c3=np.corrcoef(keep[:-7296],keep[7296:])[0,1]
    # different keys
    h2=h(key+1,idx); k2=np.stack([(h2&np.uint64(0xFFFF))>=thr,(h2>>np.uint64(16))>=thr],1).reshape(-1).astype(np.float64)
    ck=np.corrcoef(keep,k2)[0,1]
    return rate,c1,c2,c3,ck
for name,h in (("current",cur),("candA",candA),("candB",candB)):
    print(name,"avalanche max/mean dev",avalanche(h),"seq keep-rate / corr",["%.5f"%v for v in seq_stats(h)])
def candC(key,idx):   # 2 mad_u24 rounds
    x=(u(idx)+np.uint64(key))&M32
    x^=x>>np.uint64(16)
    x=(mul24(x,0xB5297B)+rotl(x,9))&M32          # v_alignbit, v_mad_u32_u24
    x^=x>>np.uint64(13)
    x=(mul24(x,0x68E31F)+rotl(x,11))&M32
    x^=x>>np.uint64(15)
    return x
def candD(key,idx):   # mul_u24 of both 16-bit halves crosswise
    x=(u(idx)+np.uint64(key))&M32
    x^=x>>np.uint64(15)
    x=(mul24(x,0x2C1B3C6D&0xFFFFFF)+rotl(x,8))&M32
    x^=x>>np.uint64(12)
    x=(mul24(x,0x297A2D39&0xFFFFFF)+rotl(x,8))&M32
    x^=x>>np.uint64(15)
    return x
def candE(key,idx):   # one 32-bit mul (quarter) + one u24
    x=(u(idx)+np.uint64(key))&M32
    x^=x>>np.uint64(16)
    x=(x*np.uint64(0x7feb352d))&M32
    x^=x>>np.uint64(15)
    x=(mul24(x,0x46ca6b)+rotl(x,10))&M32
    x^=x>>np.uint64(16)
    return x
for name,h in (("candC",candC),("candD",candD),("candE",candE)):
    print(name,"avalanche max/mean dev",avalanche(h),"seq keep-rate / corr",["%.5f"%v for v in seq_stats(h)])
