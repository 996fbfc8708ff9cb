"""ORACLE (test infrastructure, never shipped in the product path).

numpy/float64 executable specification of the third-party numerics behind the
reference's WORLD analysis path (pyworld: Dio, StoneMask, CheapTrick, D4C+LoveTrain,
CodeAperiodicity; pysptk: mcep, mgc2sp) -- restated from the published algorithms
(mmorise/World, SPTK 3.x) because neither library is in /root/reference nor installable.
Reference call sites: idiaptts/src/data_preparation/world/WorldFeatLabelGen.py:792-805,
idiaptts/src/data_preparation/audio/AudioProcessing.py:146-152,252-255.

Pinned: analyse(x, 16000, 0.97, 19, 0.58) reproduces the reference's golden fixtures
test/integration/fixtures/WORLD/cmp_mcep20/*.cmp (V/UV bit-exact, <=1 f32 ulp elsewhere);
see tests/test_oracle_golden.py.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
# Verified executable specification (numpy) of the third-party numerics behind the reference's
# analysis path: WORLD Dio + StoneMask + CheapTrick + D4C(LoveTrain) + CodeAperiodicity and SPTK mcep.
# Verified in the survey container against test/integration/fixtures/WORLD/cmp_mcep20/*.cmp
# (LJ001-0001/-0002/-0008): V/UV bit-exact, every column within 1 float32 ulp.
# Settings that reproduce the fixtures: preemphasis 0.97 (AudioProcessing.get_raw), frame_period 5 ms,
# fft_size 1024, mcep order 19, alpha 0.58, eps 1e-8, miniter 2, maxiter 30, threshold 1e-3.
# All arithmetic is float64; the reference casts mcep/lf0/vuv/bap to float32 afterwards.
import numpy as np, math
EPS = 1e-12            # WORLD kMySafeGuardMinimum
KMAX = 100000.0        # WORLD kMaximumValue
def mround(x): return int(x + 0.5) if x > 0 else int(x - 0.5)          # matlab_round

# ---------- shared helpers ----------
def interp1Q(x0, shift, y, xi):            # WORLD interp1Q: equally spaced abscissa starting at x0 with step `shift`
    y = np.asarray(y); pos = (xi - x0) / shift; base = pos.astype(np.int64); frac = pos - base
    dy = np.empty(len(y)); dy[:-1] = np.diff(y); dy[-1] = 0.0
    return y[base] + dy[base] * frac
def histc(x, edges):                       # WORLD histc (1-based bin index, clamps to [1, len(x)-1])
    n = len(x); m = len(edges); index = np.zeros(m, dtype=np.int64); count = 1; i = 0
    while i < m:
        index[i] = 1
        if edges[i] >= x[0]: break
        i += 1
    while i < m:
        if edges[i] < x[count]: index[i] = count
        else:
            index[i] = count; count += 1; i -= 1
        if count == n: break
        i += 1
    count -= 1; i += 1
    while i < m:
        index[i] = count; i += 1
    return index
def interp1(x, y, xi):                     # WORLD interp1: linear, **extrapolates** with first/last segment
    x = np.asarray(x); y = np.asarray(y); h = np.diff(x); k = histc(x, xi)
    s = (xi - x[k - 1]) / h[k - 1]; return y[k - 1] + s * (y[k] - y[k - 1])
def nuttall(n):
    t = np.arange(n) / (n - 1.0)
    return 0.355768 - 0.487396*np.cos(2*np.pi*t) + 0.144232*np.cos(4*np.pi*t) - 0.012604*np.cos(6*np.pi*t)
def dc_correction(P, f0, fs, fft):
    upper = 2 + int(f0 * fft / fs); lfa = np.arange(upper) * fs / fft
    rep = interp1Q(f0 - lfa[0], -fs / fft, P[:upper + 1], lfa[:upper - 1])   # spectrum mirrored around f0
    P = P.copy(); P[:upper - 1] += rep; return P
def linear_smoothing(P, width, fs, fft):   # rectangular smoothing of width `width` Hz via cumulative sum
    boundary = int(width * fft / fs) + 1; h = fft // 2
    mir = np.empty(h + boundary * 2 + 1)
    mir[:boundary] = P[boundary:0:-1]; mir[boundary:h + boundary] = P[:h]
    idx = np.arange(h + boundary, h + boundary * 2 + 1); mir[idx] = P[h - (idx - (h + boundary))]
    seg = np.cumsum(mir * fs / fft)
    fa = np.arange(h + 1) / fft * fs - width / 2.0
    org = -(boundary - 0.5) * fs / fft; dfi = fs / fft
    return (interp1Q(org, dfi, seg, fa + width) - interp1Q(org, dfi, seg, fa)) / width

# ---------- DIO ----------
def zero_crossing_engine(sig, n, fs):
    s = sig[:n]; ng = np.where((s[:-1] > 0.0) & (s[1:] <= 0.0))[0] + 1
    if len(ng) < 2: return None
    fine = ng - s[ng - 1] / (s[ng] - s[ng - 1])
    return (fine[:-1] + fine[1:]) / 2.0 / fs, fs / (fine[1:] - fine[:-1])      # (locations, interval-f0)
def select_best(cur, past, cands, nb, ti, ar):
    ref = (cur * 3.0 - past) / 2.0; me = abs(ref - cands[0, ti]); bf = cands[0, ti]
    for i in range(1, nb):
        ce = abs(ref - cands[i, ti])
        if ce < me: me = ce; bf = cands[i, ti]
    return 0.0 if abs(1.0 - bf / ref) > ar else bf
def fix_f0_contour(fp, nb, cands, best, T, f0_floor, ar):
    vrm = int(0.5 + 1000.0 / fp / f0_floor) * 2 + 1
    if T <= vrm: return np.zeros(T)
    base = np.zeros(T); base[vrm:T - vrm] = best[vrm:T - vrm]
    s1 = np.zeros(T)
    for i in range(vrm, T): s1[i] = base[i] if abs((base[i] - base[i-1]) / (EPS + base[i])) < ar else 0.0
    s2 = s1.copy(); c = (vrm - 1) // 2
    for i in range(c, T - c):
        if (s1[i - c:i + c + 1] == 0).any(): s2[i] = 0.0
    pos = []; neg = []
    for i in range(1, T):
        if s2[i] == 0 and s2[i-1] != 0: neg.append(i - 1)
        elif s2[i-1] == 0 and s2[i] != 0: pos.append(i)
    s3 = s2.copy()
    for i in range(len(neg)):
        limit = T - 1 if i == len(neg) - 1 else neg[i + 1]
        for j in range(neg[i], limit):
            s3[j + 1] = select_best(s3[j], s3[j - 1], cands, nb, j + 1, ar)
            if s3[j + 1] == 0: break
    s4 = s3.copy()
    for i in range(len(pos) - 1, -1, -1):
        limit = 1 if i == 0 else pos[i - 1]
        for j in range(pos[i], limit, -1):
            s4[j - 1] = select_best(s4[j], s4[j + 1], cands, nb, j - 1, ar)
            if s4[j - 1] == 0: break
    return s4
def dio(x, fs, frame_period=5.0, f0_floor=71.0, f0_ceil=800.0, channels_in_octave=2.0, allowed_range=0.1):
    # speed=1 (no decimation), as pyworld.dio defaults
    nb = 1 + int(math.log(f0_ceil / f0_floor) / math.log(2.0) * channels_in_octave)
    bnd = [f0_floor * 2.0 ** ((i + 1) / channels_in_octave) for i in range(nb)]
    xl = len(x); yl = 1 + xl; afs = float(fs)
    fft = 2 ** (int(math.log2(yl + mround(afs / 50.0) * 2 + 1 + 4 * int(1.0 + afs / bnd[0] / 2.0))) + 1)
    y = np.zeros(fft); y[:xl] = x; y[:yl] -= y[:yl].sum() / yl            # DC removal over yl = xl+1 samples
    N = mround(afs / 50.0) * 2 + 1                                          # 50 Hz low-cut filter (Hann-shaped, unit-minus)
    f = np.zeros(fft); i = np.arange(1, N + 1); w = 0.5 - 0.5 * np.cos(i * 2.0 * np.pi / (N + 1)); f[:N] = -w / w.sum()
    half = (N - 1) // 2
    for k in range(half): f[fft - half + k] = f[k]
    for k in range(N): f[k] = f[k + half]
    f[0] += 1.0
    Y = np.fft.rfft(y) * np.fft.rfft(f)
    T = int(1000.0 * xl / fs / frame_period) + 1; tp = np.arange(T) * frame_period / 1000.0
    cands = np.zeros((nb, T)); scores = np.zeros((nb, T))
    for b in range(nb):
        hal = mround(afs / bnd[b] / 2.0)
        lpf = np.zeros(fft); lpf[:hal * 4] = nuttall(hal * 4)
        sig = np.fft.irfft(Y * np.fft.rfft(lpf), n=fft)[hal * 2:hal * 2 + yl].copy()   # delay compensation
        ev = [zero_crossing_engine(sig, yl, afs)]
        sig = -sig; ev.append(zero_crossing_engine(sig, yl, afs))
        d = sig[:-1] - sig[1:]; ev.append(zero_crossing_engine(d, yl - 1, afs))
        d = -d; ev.append(zero_crossing_engine(d, yl - 1, afs))
        if any(e is None or len(e[0]) - 2 <= 0 for e in ev):
            cands[b] = 0.0; scores[b] = KMAX
        else:
            sets = np.array([interp1(e[0], e[1], tp) for e in ev])
            c = sets.sum(0) / 4.0; sc = np.sqrt(((sets - c) ** 2).sum(0) / 3.0)
            bad = (c > bnd[b]) | (c < bnd[b] / 2.0) | (c > f0_ceil) | (c < f0_floor)
            c[bad] = 0.0; sc[bad] = KMAX; cands[b] = c; scores[b] = sc
        scores[b] = scores[b] / (cands[b] + EPS)
    best = np.empty(T)
    for i in range(T):                                                     # first band wins ties (strict >)
        t = scores[0, i]; best[i] = cands[0, i]
        for j in range(1, nb):
            if t > scores[j, i]: t = scores[j, i]; best[i] = cands[j, i]
    return fix_f0_contour(frame_period, nb, cands, best, T, f0_floor, allowed_range), tp

# ---------- StoneMask (two-stage variant: 2 harmonics, then min(6, fs/2/f0) harmonics) ----------
def _fixf0(ps, num, fft, fs, f0, nh):
    amp = np.zeros(nh); inst = np.zeros(nh)
    for i in range(nh):
        idx = mround(f0 * fft / fs * (i + 1))
        inst[i] = 0.0 if ps[idx] == 0.0 else idx * fs / fft + num[idx] / ps[idx] * fs / 2.0 / np.pi
        amp[i] = math.sqrt(ps[idx])
    return (amp * inst).sum() / ((amp * (np.arange(nh) + 1)).sum() + EPS)
def stonemask_frame(x, fs, pos, f0):
    if f0 <= 40.0 or f0 > fs / 12.0: return 0.0
    half = int(1.5 * fs / f0 + 1.0); wlt = (2.0 * half + 1.0) / fs; n = 2 * half + 1
    fft = 2 ** (2 + int(math.log(half * 2.0 + 1.0) / math.log(2.0)))
    idx = mround((pos - half / fs) * fs + 0.001) + np.arange(n)
    tmp = (idx - 1.0) / fs - pos
    mw = 0.42 + 0.5 * np.cos(2.0 * np.pi * tmp / wlt) + 0.08 * np.cos(4.0 * np.pi * tmp / wlt)
    dw = np.empty(n); dw[0] = -mw[1] / 2.0; dw[1:-1] = -(mw[2:] - mw[:-2]) / 2.0; dw[-1] = mw[-2] / 2.0
    safe = np.maximum(0, np.minimum(len(x) - 1, idx - 1))
    M = np.fft.rfft(x[safe] * mw, fft); D = np.fft.rfft(x[safe] * dw, fft)
    num = M.real * D.imag - M.imag * D.real; ps = M.real ** 2 + M.imag ** 2
    t = _fixf0(ps, num, fft, fs, f0, 2)
    mean = 0.0 if (t <= 0.0 or t > f0 * 2) else _fixf0(ps, num, fft, fs, t, min(int(fs / 2.0 / f0), 6))
    return f0 if abs(mean - f0) > f0 * 0.2 else mean

# ---------- WORLD randn(): xorshift128, fixed seed restored by randn_reseed() ----------
class XorShift:
    def __init__(s): s.x,s.y,s.z,s.w=123456789,362436069,521288629,88675123
    def _step(s):
        t=(s.x^((s.x<<11)&0xFFFFFFFF))&0xFFFFFFFF; s.x,s.y,s.z=s.y,s.z,s.w
        s.w=((s.w^(s.w>>19))^(t^(t>>8)))&0xFFFFFFFF; return s.w
    def randn(s):
        tmp=s._step()>>4
        for _ in range(11): tmp+=s._step()>>4
        return tmp/268435456.0-6.0

# ---------- CheapTrick (one frame) ----------
def cheaptrick_frame(x, fs, f0, pos, fft_size, q1=-0.15, rng=None):
    # caller passes f0 = 500.0 when f0 <= 3*fs/(fft_size-3); rng = the XorShift stream of this
    # CheapTrick() call (frames consume it in order); None drops WORLD's two safeguard noise terms
    half = mround(1.5 * fs / f0); base = np.arange(-half, half + 1)
    safe = np.minimum(len(x) - 1, np.maximum(0, mround(pos * fs + 0.001) + base))
    win = 0.5 * np.cos(np.pi * (base / 1.5 / fs) * f0) + 0.5; win /= np.sqrt(np.sum(win * win))
    wf = x[safe] * win
    if rng is not None: wf = wf + np.array([rng.randn() for _ in range(len(base))]) * 1e-12
    wf = wf - win * (wf.sum() / win.sum())
    S = np.fft.rfft(wf, fft_size); P = dc_correction(S.real ** 2 + S.imag ** 2, f0, fs, fft_size)
    P = linear_smoothing(P, f0 * 2.0 / 3.0, fs, fft_size)
    if rng is not None: P = P + np.abs(np.array([rng.randn() for _ in range(fft_size // 2 + 1)])) * 2.2204460492503131e-16
    h = fft_size // 2; q = np.arange(1, h + 1) / fs
    sl = np.ones(h + 1); cl = np.ones(h + 1)
    sl[1:] = np.sin(np.pi * f0 * q) / (np.pi * f0 * q); cl[1:] = (1 - 2 * q1) + 2 * q1 * np.cos(2 * np.pi * q * f0)
    lp = np.log(P); C = np.fft.rfft(np.concatenate([lp, lp[h - 1:0:-1]])).real
    return np.exp(np.fft.irfft(C * sl * cl, n=fft_size)[:h + 1])   # power spectral envelope, fft_size/2+1 bins

# ---------- D4C + LoveTrain + CodeAperiodicity ----------
def _windowed(x, fs, f0, pos, wtype, ratio):
    half = mround(ratio * fs / f0 / 2.0); base = np.arange(-half, half + 1)
    safe = np.minimum(len(x) - 1, np.maximum(0, mround(pos * fs + 0.001) + base))
    p = (2.0 * base / ratio) / fs
    win = 0.5*np.cos(np.pi*p*f0) + 0.5 if wtype == "hanning" else 0.42 + 0.5*np.cos(np.pi*p*f0) + 0.08*np.cos(np.pi*p*f0*2)
    wf = x[safe] * win; return wf - win * (wf.sum() / win.sum())
def _centroid(x, fs, f0, fft, pos):
    wf = _windowed(x, fs, f0, pos, "blackman", 4.0); n = mround(2.0 * fs / f0) * 2 + 1
    buf = np.zeros(fft); buf[:len(wf)] = wf; buf[:n] /= np.sqrt(np.sum(buf[:n] ** 2))
    S1 = np.fft.rfft(buf); S2 = np.fft.rfft(buf * (np.arange(fft) + 1.0))
    return S2.real * S1.real + S1.imag * S2.imag
def _d4c_frame(x, fs, f0, pos, fftd, nap, window):
    wl = len(window)
    sc = dc_correction(_centroid(x, fs, f0, fftd, pos - 0.25 / f0) + _centroid(x, fs, f0, fftd, pos + 0.25 / f0), f0, fs, fftd)
    S = np.fft.rfft(_windowed(x, fs, f0, pos, "hanning", 4.0), fftd)
    sps = linear_smoothing(dc_correction(S.real ** 2 + S.imag ** 2, f0, fs, fftd), f0, fs, fftd)
    sgd = linear_smoothing(sc / sps, f0 / 2.0, fs, fftd); sgd = sgd - linear_smoothing(sgd, f0, fs, fftd)
    boundary = mround(fftd * 8.0 / wl); half = wl // 2; out = []
    for i in range(nap):
        center = int(3000.0 * (i + 1) * fftd / fs)
        S = np.fft.rfft(sgd[center - half:center + half + 1] * window, fftd)
        cs = np.cumsum(np.sort(S.real ** 2 + S.imag ** 2))
        out.append(min(0.0, 10 * np.log10(cs[fftd // 2 - boundary - 1] / cs[fftd // 2]) + (f0 - 100) / 50.0))
    return np.array(out)
def _lovetrain(x, fs, f0, pos):
    fft = 2 ** (1 + int(math.log2(3.0 * fs / 40.0 + 1)))
    b0 = int(math.ceil(100.0 * fft / fs)); b1 = int(math.ceil(4000.0 * fft / fs)); b2 = int(math.ceil(7900.0 * fft / fs))
    S = np.fft.rfft(_windowed(x, fs, max(f0, 40.0), pos, "blackman", 3.0), fft)
    ps = S.real ** 2 + S.imag ** 2; ps[:b0 + 1] = 0.0; cs = np.cumsum(ps[:b2 + 1]); return cs[b1] / cs[b2]
def d4c_bap(x, fs, f0, tp, fft_size, threshold=0.85):
    # returns code_aperiodicity(d4c(...)) directly: [T, n_bap] in dB
    T = len(f0); fftd = 2 ** (1 + int(math.log2(4.0 * fs / 47.0 + 1)))
    nap = int(min(15000.0, fs / 2.0 - 3000.0) / 3000.0)
    window = nuttall(int(3000.0 * fftd / fs) * 2 + 1)
    fa = np.arange(fft_size // 2 + 1) * fs / fft_size
    cfa = np.concatenate([np.arange(nap + 1) * 3000.0, [fs / 2.0]]); coarse_f = 3000.0 * (np.arange(nap) + 1)
    bap = np.empty((T, nap))
    for i in range(T):
        ap = np.full(fft_size // 2 + 1, 1 - EPS)
        if f0[i] != 0 and _lovetrain(x, fs, f0[i], tp[i]) > threshold:
            ca = _d4c_frame(x, fs, max(47.0, f0[i]), tp[i], fftd, nap, window)
            ap = 10 ** (np.interp(fa, cfa, np.concatenate([[-60.0], ca, [-EPS]])) / 20.0)
        bap[i] = np.interp(coarse_f, fa, 20 * np.log10(ap))
    return bap

# ---------- SPTK freqt / frqtr / mcep (itype=3 amplitude input, etype=1) ----------
def freqt(c1, m1, m2, a):
    b = 1 - a * a; g = np.zeros(m2 + 1); d = np.zeros(m2 + 1)
    for i in range(-m1, 1):
        d[0] = g[0]; g[0] = c1[-i] + a * d[0]
        if m2 >= 1: d[1] = g[1]; g[1] = b * d[0] + a * d[1]
        for j in range(2, m2 + 1): d[j] = g[j]; g[j] = d[j - 1] + a * (d[j] - g[j - 1])
    return g
def frqtr(c1, m1, m2, a):
    g = np.zeros(m2 + 1); d = np.zeros(m2 + 1)
    for i in range(-m1, 1):
        d[0] = g[0]; g[0] = c1[-i]
        for j in range(1, m2 + 1): d[j] = g[j]; g[j] = d[j - 1] + a * (d[j] - g[j - 1])
    return g
def merlin_post_filter(mgc, alpha, minimum_phase_order=511, fftlen=1024, coef=1.4):
    # nnmnkwii.postfilters.merlin_post_filter restated with SPTK's freqt / c2acr / mc2b / b2mc
    T, D = mgc.shape; w = np.ones(D) * coef; w[:2] = 1
    def c2acr0(c):                                   # 0th autocorrelation of the minimum-phase spectrum
        cc = np.zeros(fftlen); cc[:len(c)] = c
        return np.mean(np.exp(2 * np.fft.fft(cc).real))
    def mc2b(mc):
        b = mc.copy()
        for m in range(D - 2, -1, -1): b[m] = mc[m] - alpha * b[m + 1]
        return b
    out = np.zeros_like(mgc)
    for t in range(T):
        r0 = c2acr0(freqt(mgc[t], D - 1, minimum_phase_order, -alpha))
        r0p = c2acr0(freqt(mgc[t] * w, D - 1, minimum_phase_order, -alpha))
        b = mc2b(mgc[t] * w); b[0] = np.log(r0 / r0p) / 2 + b[0]
        out[t] = b; out[t, :-1] = b[:-1] + alpha * b[1:]
    return out
def sptk_mcep(amp, m, a, eps=1e-8, itr1=2, itr2=30, dd=1e-3):
    flng = (len(amp) - 1) * 2; f2 = flng // 2; m2 = 2 * m
    x = amp * amp + eps; x = np.concatenate([x, x[f2 - 1:0:-1]])
    c = np.fft.ifft(np.log(x)).real; c[0] /= 2; c[f2] /= 2
    mc = freqt(c, f2, m, a); s = c[0]; al = (-a) ** np.arange(m + 1)
    for j in range(1, itr2 + 1):
        cc = np.zeros(flng); cc[:f2 + 1] = freqt(mc, m, f2, -a)
        r = np.fft.ifft(x / np.exp(2 * np.fft.fft(cc).real)).real
        cr = frqtr(r, f2, m2, a); t = cr[0]
        if j >= itr1:
            if abs((t - s) / t) < dd: break
            s = t
        b = cr[:m + 1] - al
        hk = cr.copy(); hk[0:m2 + 1:2] -= cr[0]                 # Hankel part
        tp_ = cr.copy(); tp_[2:m + 1:2] += cr[0]; tp_[0] += cr[0]  # Toeplitz part
        A = np.array([[tp_[abs(i - k)] + hk[i + k] for k in range(m + 1)] for i in range(m + 1)])
        mc = mc + np.linalg.solve(A, b)                          # SPTK: theq()
    return mc
def mcep_to_amp_sp(mc, alpha, fftlen):                           # pysptk.mgc2sp(gamma=0) -> exp(real)
    cc = np.zeros(fftlen); cc[:fftlen // 2 + 1] = freqt(mc, len(mc) - 1, fftlen // 2, -alpha)
    return np.exp(np.fft.fft(cc).real[:fftlen // 2 + 1])

# ---------- the reference's world_extract_features + extract_mcep, restated ----------
def analyse(x_raw, fs, preemphasis, order, alpha, frame_period=5.0):
    x = np.append(x_raw[0], x_raw[1:] - preemphasis * x_raw[:-1])
    f0d, tp = dio(x, fs, frame_period)
    f0 = np.array([stonemask_frame(x, fs, tp[i], f0d[i]) for i in range(len(tp))])
    fft = 2 ** (1 + int(math.log2(3.0 * fs / 71.0 + 1))); floor = 3.0 * fs / (fft - 3.0)
    rng = XorShift()
    sp = np.array([cheaptrick_frame(x, fs, (f0[i] if f0[i] > floor else 500.0), tp[i], fft, rng=rng) for i in range(len(tp))])
    bap = d4c_bap(x, fs, f0, tp, fft)
    mc = np.array([sptk_mcep(np.sqrt(s), order, alpha) for s in sp])
    return f0, sp, bap, mc
