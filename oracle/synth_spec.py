"""ORACLE (test infrastructure, never shipped in the product path).

numpy/float64 restatement of WORLD Synthesis / DecodeAperiodicity (pyworld.synthesize,
pyworld.decode_aperiodicity; reference call site WorldFeatLabelGen.py:940-943).
PARITY UNPINNED: the reference holds no golden waveform; this restates the published
WORLD algorithm (mmorise/World synthesis.cpp) and is sanity-checked by re-analysis only.
"""
# Recollection of WORLD Synthesis / DecodeAperiodicity (numpy, float64). NOT pinned by any reference golden;
# sanity-checked only by re-analysis.
import numpy as np, math
from .world_spec import interp1, EPS, XorShift
def decode_aperiodicity(bap, fs, fft_size):
    T,n=bap.shape; fa=np.arange(fft_size//2+1)*fs/fft_size
    cfa=np.concatenate([np.arange(n+1)*3000.0,[fs/2.0]]); out=np.empty((T,fft_size//2+1))
    for i in range(T):
        out[i]=10**(interp1(cfa,np.concatenate([[-60.0],bap[i],[-EPS]]),fa)/20.0)
    return out
def _min_phase(log_half, fft):          # log_half: fft/2+1 values of log-amplitude
    full=np.concatenate([log_half, log_half[fft//2-1:0:-1]])
    cep=np.fft.ifft(full)               # real, even
    c=np.zeros(fft,dtype=complex); c[0]=cep[0]; c[1:fft//2]=2*cep[1:fft//2]; c[fft//2]=cep[fft//2]
    return np.exp(np.fft.fft(c))[:fft//2+1]
def synthesize(f0, sp, ap, fs, frame_period=5.0, time_shift=True):
    T=len(f0); fft=(sp.shape[1]-1)*2; yl=int(T*frame_period*fs/1000); fp=frame_period/1000.0
    rng=XorShift(); y=np.zeros(yl)
    lowest=fs/fft+1.0
    cta=np.arange(T+1)*fp; cf0=np.where(f0<lowest,0.0,f0); cv=(cf0!=0).astype(float)
    cf0=np.append(cf0,cf0[-1]*2-cf0[-2]); cv=np.append(cv,cv[-1]*2-cv[-2])
    ta=np.arange(yl)/fs
    if0=interp1(cta,cf0,ta); iv=(interp1(cta,cv,ta)>0.5).astype(float); if0=np.where(iv==0,500.0,if0)
    total=np.cumsum(2*np.pi*if0/fs); wrap=np.fmod(total,2*np.pi)
    idx=np.where(np.abs(wrap[1:]-wrap[:-1])>np.pi)[0]
    y1=wrap[idx]-2*np.pi; y2=wrap[idx+1]; shift=(-y1/(y2-y1))/fs if time_shift else np.zeros(len(idx))
    i=np.arange(fft//2); dcr=np.zeros(fft); dcr[:fft//2]=0.5-0.5*np.cos(2*np.pi*(i+1.0)/(1.0+fft)); dcr[fft-1-i]=dcr[i]; dcr/=dcr[:fft//2].sum()*2
    P=len(idx)
    for p in range(P):
        noise_size=idx[min(P-1,p+1)]-idx[p]; t=ta[idx[p]]
        fl=min(T-1,int(math.floor(t/fp))); ce=min(T-1,int(math.ceil(t/fp))); a=t/fp-fl
        se=np.abs(sp[fl]) if fl==ce else (1-a)*np.abs(sp[fl])+a*np.abs(sp[ce])
        sa=lambda v: np.clip(v,0.001,0.999999999999)**2
        ar=sa(ap[fl]) if fl==ce else (1-a)*sa(ap[fl])+a*sa(ap[ce])
        vuv=iv[idx[p]]
        if vuv<=0.5 or ar[0]>0.999: per=np.zeros(fft)
        else:
            mp=_min_phase(np.log(se*(1-ar)+EPS)/2.0,fft)
            if time_shift:
                k=np.arange(fft//2+1); coef=2*np.pi*shift[p]*fs/fft; mp=mp*np.exp(-1j*coef*k)   # delay by shift
            per=np.fft.fftshift(np.fft.irfft(mp,n=fft))
            dc=per[fft//2:].sum(); new=np.empty(fft); new[:fft//2]=-dc*dcr[:fft//2]; new[fft//2:]=per[fft//2:]-dc*dcr[fft//2:]; per=new
        nz=np.zeros(fft)
        if noise_size>0:
            w=np.array([rng.randn() for _ in range(noise_size)]); nz[:noise_size]=w-w.mean()
        ls=np.log(se*ar)/2.0 if vuv!=0 else np.log(se)/2.0
        ape=np.fft.fftshift(np.fft.irfft(_min_phase(ls,fft)*np.fft.rfft(nz),n=fft))
        resp=per*math.sqrt(noise_size)+ape   # WORLD divides by fft_size because its inverse FFT is unnormalised; numpy irfft already is
        off=idx[p]-fft//2+1; lo=max(0,-off); hi=min(fft,yl-off)
        y[lo+off:hi+off]+=resp[lo:hi]
    return y
