"""numpy prototype of the native overlap-save pipeline (index math only): Stockham radix-4 in
'LDS', four-step N = N1*N2, two real frames per complex transform, permuted spectrum."""
import numpy as np


def stockham4(a, inverse=False):
    """Radix-4 Stockham autosort FFT along axis 0 (length power of 4), natural order in/out."""
    n = a.shape[0]
    sgn = 1.0 if inverse else -1.0
    ns = 1
    x = a.astype(np.complex128)
    while ns < n:
        y = np.empty_like(x)
        q = n // 4
        for j in range(q):
            k = j % ns
            v = [x[j + r * q] * np.exp(sgn * 2j * np.pi * r * k / (ns * 4)) for r in range(4)]
            i = 1j * (-sgn)     # forward: multiply by -i -> here use generic DFT4
            X0 = v[0] + v[1] + v[2] + v[3]
            X1 = v[0] + (sgn * 1j) * v[1] - v[2] - (sgn * 1j) * v[3]
            X2 = v[0] - v[1] + v[2] - v[3]
            X3 = v[0] - (sgn * 1j) * v[1] - v[2] + (sgn * 1j) * v[3]
            j0 = (j // ns) * ns * 4 + k
            y[j0], y[j0 + ns], y[j0 + 2 * ns], y[j0 + 3 * ns] = X0, X1, X2, X3
        x = y
        ns *= 4
    return x


def test_stockham():
    rng = np.random.default_rng(0)
    for n in (4, 16, 64, 256):
        a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        assert np.allclose(stockham4(a), np.fft.fft(a))
        assert np.allclose(stockham4(a, True), np.fft.ifft(a) * n)


def conv_pair(xa, xb, kf, N1, N2):
    """xa, xb: two real frames of length N; kf flipped taps. Returns correlation outputs (first S valid)."""
    N = N1 * N2
    K = len(kf)
    z = xa + 1j * xb
    # spectrum of the (real) flipped kernel, correlation form: conj(FFT(kf_pad)) / N
    H = np.conj(np.fft.fft(np.pad(kf, (0, N - K)))) / N
    # permuted for pass B: Hp[k1][k2] = H[k1 + N1*k2]
    Hp = H.reshape(N2, N1).T.copy()          # H index = k2*N1 + k1 -> [k2][k1] -> transpose
    # pass A: columns n2, FFT over n1 (n = n1*N2 + n2)
    A = z.reshape(N1, N2)                    # [n1][n2]
    T1 = stockham4(A)                        # FFT over n1 (axis 0) -> [k1][n2]
    # pass B: twiddle, row FFT over n2, *H, row IFFT, conj twiddle
    k1 = np.arange(N1)[:, None]
    n2 = np.arange(N2)[None, :]
    tw = np.exp(-2j * np.pi * k1 * n2 / N)
    B = T1 * tw
    B = stockham4(B.T).T                     # FFT over n2 -> [k1][k2]
    B = B * Hp
    B = stockham4(B.T, inverse=True).T       # IFFT over k2 -> [k1][n2]
    B = B * np.conj(tw)
    # pass C: IFFT over k1 -> y[n1*N2 + n2]
    Y = stockham4(B, inverse=True)           # [n1][n2]
    y = Y.reshape(-1)
    return y.real, y.imag


def test_conv():
    rng = np.random.default_rng(1)
    N1, N2 = 16, 64
    N = N1 * N2
    K = 200
    kf = rng.standard_normal(K)
    xa, xb = rng.standard_normal(N), rng.standard_normal(N)
    ya, yb = conv_pair(xa, xb, kf, N1, N2)
    S = N - K + 1
    for x, y in ((xa, ya), (xb, yb)):
        ref = np.array([np.dot(kf, x[i:i + K]) for i in range(S)])
        assert np.allclose(y[:S], ref, atol=1e-9), np.abs(y[:S] - ref).max()


if __name__ == "__main__":
    test_stockham()
    test_conv()
    print("prototype ok")
