"""Numerical model of csrc/dft_mfma.hip: 2-D DFT -> influence-like weighting -> inverse 2-D DFT of a 75 x 75 plane by matrix
products whose operands are split x = hi + lo into two f16 (hi = x truncated to 11 significant bits, lo = the residual,
both as real f16 incl. subnormals), product = hi*hi + lo*hi + hi*lo in ONE f32 accumulator, plane scaled to 2^13 at load
and by the a-priori factor 2^-5 after every pass.  Compared with an f64 reference and with an f32 FFT."""
import numpy as np
n = 75
k = np.arange(n)
W = np.exp(-2j * np.pi * np.outer(k, k) / n)

def split(x):
    x = x.astype(np.float32)
    h = (x.view(np.uint32) & np.uint32(0xffffe000)).view(np.float32)
    hi = h.astype(np.float16).astype(np.float32)
    lo = (x - h).astype(np.float16).astype(np.float32)
    return hi, lo

def mm(a, b):
    return (a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)     # exact products, f32 result

def cmm(Ar, Ai, Wr, Wi, inv):
    arh, arl = split(Ar); aih, ail = split(Ai); wrh, wrl = split(Wr); wih, wil = split(Wi)
    def prod(ah, al, wh, wl): return mm(ah, wh) + mm(al, wh) + mm(ah, wl)
    p1 = prod(arh, arl, wrh, wrl); p2 = prod(aih, ail, wih, wil); q1 = prod(arh, arl, wih, wil); q2 = prod(aih, ail, wrh, wrl)
    return (p1 + p2, q2 - q1) if inv else (p1 - p2, q1 + q2)

def pipeline(A, G):
    mx = max(np.abs(A.real).max(), np.abs(A.imag).max())
    s = 2.0 ** (13 - np.floor(np.log2(mx)))
    gb = 2.0 ** np.ceil(np.log2(G.max()))
    Wr, Wi = W.real.astype(np.float32), W.imag.astype(np.float32)
    d_r, d_i = (A.real * s).astype(np.float32), (A.imag * s).astype(np.float32)
    sc = np.float32(2.0 ** -5)
    r, i = cmm(d_r, d_i, Wr, Wi, False); d_r, d_i = (r * sc).T.copy(), (i * sc).T.copy()          # [ky][x]
    r, i = cmm(d_r, d_i, Wr, Wi, False); g = (G.T / gb).astype(np.float32)                         # D[m=ky][n=kx] -> stored [kx][ky]
    d_r, d_i = (r * sc * g).T.copy(), (i * sc * g).T.copy()
    r, i = cmm(d_r, d_i, Wr, Wi, True); d_r, d_i = (r * sc).T.copy(), (i * sc).T.copy()
    r, i = cmm(d_r, d_i, Wr, Wi, True)
    out = (r + 1j * i).T * (32.0 ** 3) * gb / s
    return out, [np.abs(d_r).max()]

def run_model(seed=0):
    """max / rms relative error of the split-f16 pipeline and of an f32 FFT on a mesh-like plane."""
    rng = np.random.default_rng(seed)
    # mesh-like plane and an influence-like weight spanning many decades
    A = np.zeros((n, n))
    for _ in range(600):
        i, j = rng.integers(0, n, 2); q = rng.choice([-0.834, 0.417, 0.417])
        w = rng.random(5); w /= w.sum(); v = rng.random(5); v /= v.sum()
        A[np.ix_((i + np.arange(5)) % n, (j + np.arange(5)) % n)] += q * np.outer(w, v)
    A = A + 1j * np.roll(A, 3, 0) * 0.7
    m = np.minimum(k, n - k)[:, None] ** 2 + np.minimum(k, n - k)[None, :] ** 2 + 4.0
    G = np.exp(-0.02 * m) / m
    ref = np.fft.ifft2(np.fft.fft2(A) * G) * n * n
    got, info = pipeline(A, G)
    f32 = np.fft.ifft2((np.fft.fft2(A.astype(np.complex64)) * G.astype(np.float32)).astype(np.complex64)).astype(np.complex128) * n * n
    err = lambda x: (np.abs(x - ref).max() / np.abs(ref).max(), np.sqrt((np.abs(x - ref) ** 2).mean() / (np.abs(ref) ** 2).mean()))

    return err(got), err(f32), info


if __name__ == '__main__':
    e_split, e_f32, info = run_model()
    print('split-f16 pipeline: max / rms relative error', e_split, ' last pass input max', info)
    print('numpy (f64 internally, f32 in/out)        ', e_f32)
