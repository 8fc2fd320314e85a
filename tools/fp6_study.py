"""Emulation of fc0 with fp6 (e2m3, block-scaled) correction terms vs the fp8 scheme in use (DESIGN.md 3.1, next lever)."""
import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
import omok_ai_amd as oa
from oracle import train as OT
torch.set_num_threads(8)
def q_e4m3(x):  # OCP e4m3fn RNE, saturating
    x=x.clone(); s=torch.sign(x); a=x.abs().clamp(max=448.0)
    e=torch.floor(torch.log2(a.clamp(min=1e-300))).clamp(min=-6.0)
    step=torch.pow(2.0,e-3)
    return s*torch.round(a/step)*step
def q_e2m3(x):  # fp6 e2m3 RNE saturating: normals 1..7.5, subnormal step .125
    s=torch.sign(x); a=x.abs().clamp(max=7.5)
    e=torch.floor(torch.log2(a.clamp(min=1e-300))).clamp(min=0.0)
    step=torch.pow(2.0,e-3)
    return s*torch.round(a/step)*step
def blockq(x, dim_groups, elem_q, emax):
    # x [..., G, 32]: per-block power-of-two scale from the block max (OCP MX: floor(log2 max) - emax)
    m=x.abs().amax(dim=-1,keepdim=True)
    se=torch.floor(torch.log2(m.clamp(min=1e-300)))-emax
    sc=torch.pow(2.0,se)
    return elem_q(x/sc)*sc
def run(n,seed,B=64):
    tensors=[np.asarray(t,np.float64) for t in oa.weights.init_random(n,seed=seed)]
    rng=np.random.default_rng(3)
    from oracle import oracle as O
    xs=[]
    for _ in range(B):
        env=O.Environment(n)
        for c in rng.permutation(n*n)[:int(rng.integers(0,n*n-1))]: env.place_stone(int(c))
        xs.append(env.encode_nn_input(int(rng.integers(0,2))))
    x=np.stack(xs).reshape(B,n,n,3)
    hw=n*n
    # trunk in f64 up to fc0 input
    t=[np.asarray(a,np.float64).reshape(s) for a,s in zip(tensors,OT.shapes(n))]
    a=np.asarray(x,np.float64); a=OT._lrelu(a@t[0][0,0]+t[1])
    for i in range(3):
        w0,b0,dw,pw,b1,w2,b2=t[2+7*i:9+7*i]
        h=OT._lrelu(a@w0[0,0]+b0); hp=np.pad(h,((0,0),(1,1),(1,1),(0,0))); d=np.zeros_like(h)
        for dy in range(3):
            for dx in range(3): d+=hp[:,dy:dy+n,dx:dx+n,:]*dw[dy,dx,:,0]
        g=OT._lrelu(d@pw[0,0]+b1); a=OT._lrelu((g@w2[0,0]+b2)+a)
    A=torch.from_numpy(a.reshape(B,hw,128)).float().double()  # f32 trunk output
    W=torch.from_numpy(t[23]).double().reshape(hw,128,512)
    def tail(h0pre):
        h0=OT._lrelu(h0pre.numpy()+t[24]); h1=OT._lrelu(h0@t[25]+t[26])
        v=np.tanh(h1@t[27]+t[28]); lg=h1@t[29]+t[30]; e=np.exp(lg-lg.max(1,keepdims=True)); return e/e.sum(1,keepdims=True), v
    ref=torch.einsum('bpc,pco->bo',A,W)
    p64,v64=tail(ref)
    Ah=A.half().double(); Al=A-Ah; Wh=W.half().double(); Wl=W-Wh
    main=torch.einsum('bpc,pco->bo',Ah,Wh)
    # channel grouping into MX blocks: (q = c>>6, h = (c>>2)&1) -> 32 channels
    c=torch.arange(128); grp=((c>>6)*2+((c>>2)&1))
    order=torch.argsort(grp,stable=True)  # [128] grouped 4 x 32
    def groupA(X): return X[:,:,order].reshape(B,hw,4,32)
    def ungroupA(Xg):
        out=torch.empty(B,hw,128,dtype=torch.float64); out[:,:,order]=Xg.reshape(B,hw,128); return out
    def groupW(X): return X[:,order,:].reshape(hw,4,32,512).permute(0,1,3,2)  # [hw,4,512,32]
    def ungroupW(Xg):
        out=torch.empty(hw,128,512,dtype=torch.float64); out[:,order,:]=Xg.permute(0,1,3,2).reshape(hw,128,512); return out
    res={}
    # (a) current: fp8 with global scales
    SA=2; sw=int(np.floor(np.log2(448.0/float(Wh.abs().max()))))
    Ah8=q_e4m3(Ah*2.0**SA)/2.0**SA; Al8=q_e4m3(Al*2.0**(SA+11))/2.0**(SA+11)
    Wh8=q_e4m3(Wh*2.0**sw)/2.0**sw; Wl8=q_e4m3(Wl*2.0**(sw+11))/2.0**(sw+11)
    corr=torch.einsum('bpc,pco->bo',Al8,Wh8)+torch.einsum('bpc,pco->bo',Ah8,Wl8)
    res['fp8 global scale']=main+corr
    # (b) fp6 e2m3 block-scaled
    f=lambda X,g,u: u(blockq(g(X),None,q_e2m3,2))
    Ah6=f(Ah,groupA,ungroupA); Al6=f(Al,groupA,ungroupA); Wh6=f(Wh,groupW,ungroupW); Wl6=f(Wl,groupW,ungroupW)
    res['fp6 e2m3 block scale']=main+torch.einsum('bpc,pco->bo',Al6,Wh6)+torch.einsum('bpc,pco->bo',Ah6,Wl6)
    # (c) fp8 block-scaled (for reference)
    f8=lambda X,g,u: u(blockq(g(X),None,q_e4m3,8))
    res['fp8 block scale']=main+torch.einsum('bpc,pco->bo',f8(Al,groupA,ungroupA),f8(Wh,groupW,ungroupW))+torch.einsum('bpc,pco->bo',f8(Ah,groupA,ungroupA),f8(Wl,groupW,ungroupW))
    res['no correction']=main
    for k,vv in res.items():
        p,v=tail(vv)
        print(f"n={n} seed={seed} {k:24s} max|dp|={np.abs(p-p64).max():.3e} max|dv|={np.abs(v-v64).max():.3e}  rms pre-act err={float((vv-ref).pow(2).mean().sqrt()):.3e}")
for n in (9,15):
    for seed in (0,1): run(n,seed,B=48 if n==15 else 96)
