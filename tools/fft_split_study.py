"""CPU study (round 6): would the front-end FFT on the matrix pipe be accurate enough?  A two-stage radix-16 DFT with operands split into
hi + lo fp16 terms (three exact-product MFMA terms per tile, fp32 accumulation, per-frame block exponent) against an fp32 FFT (pocketfft),
both scored in mel dB against a float64 FFT.  Result: the same accuracy -- and not built, the conversions cost what the butterflies do
(DESIGN_LOG.md, round 6).    python tools/fft_split_study.py"""
import numpy as np, scipy.fft as sfft
rng=np.random.default_rng(0)
N=512
win=(0.5-0.5*np.cos(2*np.pi*np.arange(N)/N)).astype(np.float32)
# HTK mel fb 64 bins 0..8000 for 257 bins
def melfb():
    f=np.linspace(0,8000,257); hz2mel=lambda f:2595*np.log10(1+f/700); mel2hz=lambda m:700*(10**(m/2595)-1)
    pts=mel2hz(np.linspace(hz2mel(0),hz2mel(8000),66)); fb=np.zeros((257,64))
    for m in range(64):
        lo,c,hi=pts[m],pts[m+1],pts[m+2]
        fb[:,m]=np.maximum(0,np.minimum((f-lo)/(c-lo),(hi-f)/(hi-c)))
    return fb
FB=melfb()
def f16split(v):
    hi=v.astype(np.float16); lo=(v-hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)
def W16(sign=-1):
    k=np.arange(16); return np.exp(sign*2j*np.pi*np.outer(k,k)/16)
def realmat(Wc):          # [[Wr,-Wi],[Wi,Wr]] 32x32
    return np.block([[Wc.real,-Wc.imag],[Wc.imag,Wc.real]])
WM=realmat(W16()).astype(np.float32)          # entries exactly representable? no: split
WMh,WMl=f16split(WM)
def mfma_split(Wh,Wl,X32):    # X32 float32 [32, cols]; returns float32 [32, cols]; three exact-sum terms, each rounded into fp32 acc
    xh,xl=f16split(X32)
    acc=np.zeros(X32.shape,np.float32)
    for A,B in ((Wh,xh),(Wh,xl),(Wl,xh)):
        acc=(acc.astype(np.float64)+A@B).astype(np.float32)
    return acc
def fft_split(v, scale_exp=None):
    # v float32 windowed frame; block exponent: scale so max|v| in [2^13,2^14)
    if scale_exp is None:
        m=np.abs(v).max(); scale_exp=0 if m==0 else 10-int(np.floor(np.log2(m)))
    vs=np.ldexp(v,scale_exp).astype(np.float32)
    z=vs[0::2]+1j*vs[1::2]                      # 256 complex
    Z=z.reshape(16,16)                          # [n1,n2]
    X=np.vstack([Z.real,Z.imag]).astype(np.float32)   # [32 (n1 re;im), 16 n2]
    Y=mfma_split(WMh,WMl,X)                     # [32 (k1 re;im), n2]
    Yc=(Y[:16].astype(np.float64)+1j*Y[16:].astype(np.float64))
    tw=np.exp(-2j*np.pi*np.outer(np.arange(16),np.arange(16))/256)   # [k1,n2]
    # twiddle in fp32 complex arithmetic
    a=Yc.real.astype(np.float32); b=Yc.imag.astype(np.float32); c=tw.real.astype(np.float32); d=tw.imag.astype(np.float32)
    re=(a*c-b*d).astype(np.float32); im=(a*d+b*c).astype(np.float32)
    # stage 2: DFT over n2 for each k1: columns = k1, K = n2
    X2=np.vstack([re.T,im.T]).astype(np.float32)      # [32 (n2 re;im), 16 k1]
    Y2=mfma_split(WMh,WMl,X2)                   # [32 (k2), k1]
    Zk=(Y2[:16].astype(np.float64)+1j*Y2[16:].astype(np.float64))   # [k2,k1] -> k = k1+16 k2
    Zf=np.zeros(256,complex)
    for k2 in range(16):
        for k1 in range(16): Zf[k1+16*k2]=Zk[k2,k1]
    # real split in fp32-ish (do in float64 then round to f32: generous) 
    k=np.arange(257); Zk_=Zf[k%256]; Zc=np.conj(Zf[(256-k)%256])
    Xk=0.5*(Zk_+Zc)-0.5j*np.exp(-2j*np.pi*k/512)*(Zk_-Zc)
    Xk=Xk.astype(np.complex64).astype(complex)
    return Xk*2.0**(-scale_exp)
def meldb(X):
    p=(np.abs(X)**2)@FB; return 10*np.log10(np.maximum(p,1e-10))
def fft32(v):
    return sfft.rfft(v.astype(np.float32)).astype(complex)
def study(name,x):
    errs=[];errs32=[]
    for st in range(0,len(x)-N,160):
        v=(x[st:st+N].astype(np.float32)*win).astype(np.float32)
        T=np.fft.rfft(v.astype(np.float64)); 
        dT=meldb(T); keep=dT>dT.max()-100
        errs.append(np.abs(meldb(fft_split(v))-dT)[keep]); errs32.append(np.abs(meldb(fft32(v))-dT)[keep])
    e=np.concatenate(errs); e32=np.concatenate(errs32)
    print(f'{name:28s} split: max {e.max():.2e} rms {np.sqrt((e**2).mean()):.2e} dB | fp32 FFT: max {e32.max():.2e} rms {np.sqrt((e32**2).mean()):.2e} dB')
t=np.arange(16000)/16000
study('white noise 0.5',rng.uniform(-.5,.5,16000))
study('white noise -60dB',rng.uniform(-.5,.5,16000)*1e-3)
study('tone 440',0.9*np.sin(2*np.pi*440*t))
study('bin-edge tone pair',0.4*np.sin(2*np.pi*(31.25*20.5)*t)+0.4*np.sin(2*np.pi*(31.25*57.5)*t))
study('tone + weak noise',0.9*np.sin(2*np.pi*1000*t)+1e-4*rng.standard_normal(16000))
study('harmonic stack -40dB',1e-2*sum(np.sin(2*np.pi*150*h*t)/h for h in range(1,30)))
