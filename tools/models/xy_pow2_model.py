# thread-level model of the power-of-two plane pass (index scheme + LDS bank conflicts)
import numpy as np, sys
def run(N, R1, R2=8, check_banks=True):
    T = N * N // R1; PS = N + 8; MM = R1 // R2
    rng = np.random.default_rng(0)
    P = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))   # [x][y]
    G = rng.standard_normal((N, N))
    W = lambda n, p: np.exp(-2j * np.pi * p / n)
    lds = np.zeros(N * PS, complex)
    conflicts = []
    def bank_check(addrs, what):
        if 'write' in what:
            worst = 1
            for g in range(4):
                a = addrs[16 * g: 16 * g + 16]; banks = {}
                for x in a:
                    for d in (2 * x, 2 * x + 1): banks.setdefault(d % 32, set()).add(d)
                worst = max(worst, max(len(v) for v in banks.values()))
            if worst > 1: conflicts.append((what, worst))
            return
        return bank_check_r(addrs, what)
    def bank_check_r(addrs, what):       # addrs: per-lane float2 index for one wave instruction (64 lanes); b64: two phases of 32 lanes
        worst = 1
        for ph in range(2):
            a = addrs[32 * ph: 32 * ph + 32]
            banks = {}
            for x in a:
                for d in (2 * x, 2 * x + 1):
                    banks.setdefault(d % 64, set()).add(d)
            worst = max(worst, max(len(v) for v in banks.values()))
        if worst > 1: conflicts.append((what, worst))
    def dft(v, sign):                   # natural-order DFT of the last axis
        n = len(v); k = np.arange(n)
        return np.array([np.sum(v * np.exp(sign * 2j * np.pi * k * kk / n)) for kk in range(n)])
    t = np.arange(T)
    # phase A
    y = t % N; x0 = t // N
    reg = np.zeros((T, R1), complex)
    for tt in range(T):
        v = np.array([P[x0[tt] + R2 * r, y[tt]] for r in range(R1)])
        A = dft(v, -1) * np.array([W(N, x0[tt] * k1) for k1 in range(R1)])
        reg[tt] = A
    for k1 in range(R1):
        addr = (k1 * R2 + x0) * N + y
        lds[addr] = reg[:, k1]
        if check_banks: bank_check(addr[:64], 'A write'); bank_check(addr[64:128], 'A write')
    w = np.zeros((T, R1), complex); kxr = np.zeros((T, R1), int)
    for m in range(MM):
        k1 = x0 + R2 * m
        u = np.zeros((T, R2), complex)
        for xp in range(R2):
            addr = (k1 * R2 + xp) * N + y
            u[:, xp] = lds[addr]
            if check_banks: bank_check(addr[:64], 'A read')
        for tt in range(T):
            w[tt, m * R2:(m + 1) * R2] = dft(u[tt], -1)
        for k2 in range(R2): kxr[:, m * R2 + k2] = k1 + R1 * k2
    # check fwd x
    ref = np.fft.fft(P, axis=0)
    for tt in range(0, T, 37):
        for i in range(R1): assert abs(w[tt, i] - ref[kxr[tt, i], y[tt]]) < 1e-9
    # phase B: transpose
    lds[:] = 0
    for i in range(R1):
        addr = kxr[:, i] * PS + y
        lds[addr] = w[:, i]
        if check_banks: bank_check(addr[:64], 'B write')
    kxl = t // R2; j = t % R2
    v = np.zeros((T, R1), complex)
    for r in range(R1):
        addr = kxl * PS + j + R2 * r
        v[:, r] = lds[addr]
        if check_banks: bank_check(addr[:64], 'B read'); bank_check(addr[64:128], 'B read')
    # phase C: fwd y
    def yfft(v, sign):
        Bv = np.zeros_like(v)
        for tt in range(T):
            Bv[tt] = dft(v[tt], sign) * np.array([np.exp(sign * 2j * np.pi * j[tt] * k1 / N) for k1 in range(R1)])
        for k1 in range(R1):
            addr = kxl * PS + k1 * R2 + ((j + k1) % R2)
            lds[addr] = Bv[:, k1]
            if check_banks: bank_check(addr[:64], 'C write'); bank_check(addr[64:128], 'C write')
        out = np.zeros_like(v); kk = np.zeros((T, R1), int)
        for m in range(MM):
            k1 = j + R2 * m
            u = np.zeros((T, R2), complex)
            for jp in range(R2):
                addr = kxl * PS + k1 * R2 + ((jp + k1) % R2)
                u[:, jp] = lds[addr]
                if check_banks: bank_check(addr[:64], 'C read'); bank_check(addr[64:128], 'C read')
            for tt in range(T): out[tt, m * R2:(m + 1) * R2] = dft(u[tt], sign)
            for k2 in range(R2): kk[:, m * R2 + k2] = k1 + R1 * k2
        return out, kk
    Y, kyr = yfft(v, -1)
    ref2 = np.fft.fft2(P)
    for tt in range(0, T, 41):
        for i in range(R1): assert abs(Y[tt, i] - ref2[kxl[tt], kyr[tt, i]]) < 1e-8
    # phase D: influence
    for i in range(R1): Y[:, i] *= G[kxl, kyr[:, i]]
    # phase E: inverse y.  register (m, k2) is DIF input r = m + MM * k2
    vin = np.zeros_like(Y)
    for m in range(MM):
        for k2 in range(R2): vin[:, m + MM * k2] = Y[:, m * R2 + k2]
    Z, yr = yfft(vin, +1)
    # phase F: transpose back
    for i in range(R1):
        addr = kxl * PS + yr[:, i]
        lds[addr] = Z[:, i]
        if check_banks: bank_check(addr[:64], 'F write'); bank_check(addr[64:128], 'F write')
    v = np.zeros((T, R1), complex)
    for r in range(R1):
        addr = (x0 + R2 * r) * PS + y
        v[:, r] = lds[addr]
    # phase G: inverse x
    regs = np.zeros((T, R1), complex)
    for tt in range(T):
        regs[tt] = dft(v[tt], +1) * np.array([np.conj(W(N, x0[tt] * k1)) for k1 in range(R1)])
    ldsE = np.zeros(N * N, complex)
    for k1 in range(R1): ldsE[(k1 * R2 + x0) * N + y] = regs[:, k1]
    out = np.zeros((N, N), complex)
    for m in range(MM):
        k1 = x0 + R2 * m
        u = np.zeros((T, R2), complex)
        for xp in range(R2): u[:, xp] = ldsE[(k1 * R2 + xp) * N + y]
        for tt in range(T):
            o = dft(u[tt], +1)
            for k2 in range(R2): out[k1[tt] + R1 * k2, y[tt]] = o[k2]
    want = np.fft.ifft2(np.fft.fft2(P) * G) * N * N
    print(N, 'max err', np.abs(out - want).max(), 'conflicts', sorted(set(conflicts)))
run(64, 8)
run(128, 16)
