#!/usr/bin/env python
"""CPU study for DESIGN.md section 8: would Winograd F(2x2, 3x3) on top of the split-fp16 products still be fp32 grade?
(numpy; 128 -> 64 channels, 32 x 32 output, swish-distributed activations; relative L2 error against an fp64 direct
convolution).  Result of this script:
    fp32 direct                                   2.1e-7
    split-fp16 direct (the shipped arithmetic)    0.9e-7
    Winograd, fp32 transforms, fp32 products      3.5e-7
    Winograd, fp32 transforms, split products     1.6e-7      <- 2.25x fewer MFMA products, still below fp32 direct
    Winograd, fp32 transforms, exact products     1.1e-7
    Winograd, fp64 transforms, exact products     0.7e-7      (the fp32 rounding of the transformed weights)
Not product code, not built."""
import numpy as np
rng=np.random.default_rng(0)
C,K,H=128,64,34          # input padded 34x34 -> output 32x32
x=rng.standard_normal((C,H,H)); x=(x/(1+np.exp(-x))).astype(np.float32)
w=(rng.standard_normal((K,C,3,3))*(1/(3*C**0.5))).astype(np.float32)
# fp64 direct reference
def direct(xd,wd,dt):
    out=np.zeros((K,32,32),dt)
    for ky in range(3):
        for kx in range(3):
            out+=np.einsum('kc,chw->khw',wd[:,:,ky,kx].astype(dt),xd[:,ky:ky+32,kx:kx+32].astype(dt))
    return out
ref=direct(x,w,np.float64)
rel=lambda y: np.linalg.norm(y.astype(np.float64)-ref)/np.linalg.norm(ref)
print('fp32 direct (numpy einsum fp32):', rel(direct(x,w,np.float32)))
def split(v):
    hi=v.astype(np.float16); lo=(v-hi.astype(np.float32)).astype(np.float16); return hi.astype(np.float64),lo.astype(np.float64)
def mm_split(a,b):   # a [M,Kc] fp32, b [Kc,N] fp32 -> fp32-grade product via 3 terms (exact accumulate, round to fp32 at end)
    ah,al=split(a); bh,bl=split(b)
    return (ah@bh+ah@bl+al@bh).astype(np.float32)
# split direct
s=2.0**(13-np.floor(np.log2(np.abs(w).max())))
out=np.zeros((K,32*32),np.float64)
for ky in range(3):
    for kx in range(3):
        A=x[:,ky:ky+32,kx:kx+32].reshape(C,-1).T   # [px, C]
        out+=mm_split(A,(w[:,:,ky,kx]*np.float32(s)).T).T.astype(np.float64)/s
print('split direct:', rel(out.reshape(K,32,32).astype(np.float32)))
# Winograd F(2x2,3x3)
Bt=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],np.float64)
G=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],np.float64)
At=np.array([[1,1,1,0],[0,1,-1,-1]],np.float64)
def wino(dt_tr, prod):
    # weights transform in fp64 then cast to fp32 (offline), input transform in dt_tr
    U=np.einsum('ij,kcjl,ml->kcim',G,w.astype(np.float64),G).astype(np.float32)     # [K,C,4,4]
    tiles=16
    V=np.zeros((C,tiles,tiles,4,4),dt_tr)
    for ty in range(tiles):
        for tx in range(tiles):
            d=x[:,2*ty:2*ty+4,2*tx:2*tx+4].astype(dt_tr)
            V[:,ty,tx]=np.einsum('ij,cjl,ml->cim',Bt.astype(dt_tr),d,Bt.astype(dt_tr))
    M=np.zeros((K,tiles,tiles,4,4),np.float32)
    for i in range(4):
        for j in range(4):
            a=V[:,:,:,i,j].reshape(C,-1).T.astype(np.float32)      # [tiles^2, C]
            b=U[:,:,i,j].T                                          # [C,K]
            if prod=='split':
                sc=2.0**(13-np.floor(np.log2(np.abs(b).max())))
                m=mm_split(a,(b*np.float32(sc))).astype(np.float64)/sc
            elif prod=='f32':
                m=(a@b)
            else:
                m=a.astype(np.float64)@b.astype(np.float64)
            M[:,:,:,i,j]=m.T.reshape(K,tiles,tiles).astype(np.float32)
    Y=np.einsum('ij,ktxjl,ml->ktxim',At.astype(dt_tr),M.astype(dt_tr),At.astype(dt_tr))   # [K,ty,tx,2,2]
    return Y.transpose(0,1,3,2,4).reshape(K,32,32)
print('winograd fp32 transforms + fp32 products :', rel(wino(np.float32,'f32')))
print('winograd fp32 transforms + split products:', rel(wino(np.float32,'split')))
print('winograd fp32 transforms + exact products:', rel(wino(np.float32,'f64')))
print('winograd fp64 transforms + exact products:', rel(wino(np.float64,'f64')))
