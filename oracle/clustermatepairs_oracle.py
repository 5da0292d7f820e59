"""CPU restatement of the reference's `clustermatepairs` tool.  TEST INFRASTRUCTURE ONLY.

  compact stream / fragment grouping   tools/AlignmentStream.cpp:156-221
  CheckConcordant, AddBinPairs, ...     tools/clustermatepairs.cpp:146-375, main :389-589
  MatePairEM                            tools/MatePairEM.cpp:43-636
  kmns / optra / qtran (AS 136)         tools/asa136.C
  r8_normal_01_cdf_inverse (AS 241)     tools/asa241.C:424-563

Python floats are IEEE doubles and every arithmetic step below is written in the reference's order
(pow(x, 2.0) is x*x, as GCC folds it); exp/log/sqrt are the platform libm's, like the reference's.
Orders the reference leaves to boost::unordered_* / std::sort ties are fixed to the canonical order
of SURVEY.md 8(c): maps iterate by ascending key, sorts are key-descending then index-ascending.

Parity status: UNPINNED against the reference for this tool (no golden vector exists; the reference
cannot be built here).  The one recorded fact — 20 synthetic fragments on two loci give "Created 2
clusters", 40 lines — is reproduced by tests/test_clustermatepairs.py.
"""
import math

PLUS, MINUS = 0, 1
DBL_MAX = 1.7976931348623157e308
R8_HUGE = 1.0e30


def cdiv(a, b):
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


# ------------------------------------------------------------------------------------------
# AS 241 (tools/asa241.C:424-563)
# ------------------------------------------------------------------------------------------
_A = [3.3871328727963666080, 1.3314166789178437745e+2, 1.9715909503065514427e+3, 1.3731693765509461125e+4,
      4.5921953931549871457e+4, 6.7265770927008700853e+4, 3.3430575583588128105e+4, 2.5090809287301226727e+3]
_B = [1.0, 4.2313330701600911252e+1, 6.8718700749205790830e+2, 5.3941960214247511077e+3,
      2.1213794301586595867e+4, 3.9307895800092710610e+4, 2.8729085735721942674e+4, 5.2264952788528545610e+3]
_C = [1.42343711074968357734, 4.63033784615654529590, 5.76949722146069140550, 3.64784832476320460504,
      1.27045825245236838258, 2.41780725177450611770e-1, 2.27238449892691845833e-2, 7.74545014278341407640e-4]
_D = [1.0, 2.05319162663775882187, 1.67638483018380384940, 6.89767334985100004550e-1,
      1.48103976427480074590e-1, 1.51986665636164571966e-2, 5.47593808499534494600e-4, 1.05075007164441684324e-9]
_E = [6.65790464350110377720, 5.46378491116411436990, 1.78482653991729133580, 2.96560571828504891230e-1,
      2.65321895265761230930e-2, 1.24266094738807843860e-3, 2.71155556874348757815e-5, 2.01033439929228813265e-7]
_F = [1.0, 5.99832206555887937690e-1, 1.36929880922735805310e-1, 1.48753612908506148525e-2,
      7.86869131145613259100e-4, 1.84631831751005468180e-5, 1.42151175831644588870e-7, 2.04426310338993978564e-15]


def _poly(a, x):
    v = 0.0
    for i in range(len(a) - 1, -1, -1):
        v = v * x + a[i]
    return v


def normal_01_cdf_inverse(p):
    if p <= 0.0:
        return -R8_HUGE
    if 1.0 <= p:
        return R8_HUGE
    q = p - 0.5
    if abs(q) <= 0.425:
        r = 0.180625 - q * q
        return q * _poly(_A, r) / _poly(_B, r)
    r = p if q < 0.0 else 1.0 - p
    if r <= 0.0:
        raise SystemExit(1)
    r = math.sqrt(-math.log(r))
    if r <= 5.0:
        r = r - 1.6
        value = _poly(_C, r) / _poly(_D, r)
    else:
        r = r - 5.0
        value = _poly(_E, r) / _poly(_F, r)
    return -value if q < 0.0 else value


def normalpdf(x, mu, sigma):   # tools/Common.cpp:61-69
    coeff = 1.0 / (sigma * math.sqrt(2 * math.pi))
    dist = (x - mu) / sigma
    return coeff * math.exp(-0.5 * dist * dist)


# ------------------------------------------------------------------------------------------
# AS 136 (tools/asa136.C): a[i + j*m], c[l + j*k] column-major, cluster ids 1-based
# ------------------------------------------------------------------------------------------
def kmns(a, m, n, c, k, iters):
    ic1, nc, wss = [0] * m, [0] * k, [0.0] * k
    if k <= 1 or m <= k:
        return ic1, nc, wss, 3
    ic2 = [0] * m
    an1, an2, ncp, d, itran, live = [0.0] * k, [0.0] * k, [0] * k, [0.0] * m, [0] * k, [0] * k
    for i in range(1, m + 1):
        ic1[i - 1], ic2[i - 1] = 1, 2
        dt = [0.0, 0.0]
        for il in (1, 2):
            for j in range(1, n + 1):
                da = a[i - 1 + (j - 1) * m] - c[il - 1 + (j - 1) * k]
                dt[il - 1] = dt[il - 1] + da * da
        if dt[1] < dt[0]:
            ic1[i - 1], ic2[i - 1] = 2, 1
            dt[0], dt[1] = dt[1], dt[0]
        for l in range(3, k + 1):
            db = 0.0
            for j in range(1, n + 1):
                dc = a[i - 1 + (j - 1) * m] - c[l - 1 + (j - 1) * k]
                db = db + dc * dc
            if db < dt[1]:
                if dt[0] <= db:
                    dt[1] = db
                    ic2[i - 1] = l
                else:
                    dt[1] = dt[0]
                    ic2[i - 1] = ic1[i - 1]
                    dt[0] = db
                    ic1[i - 1] = l
    for l in range(1, k + 1):
        nc[l - 1] = 0
        for j in range(1, n + 1):
            c[l - 1 + (j - 1) * k] = 0.0
    for i in range(1, m + 1):
        l = ic1[i - 1]
        nc[l - 1] += 1
        for j in range(1, n + 1):
            c[l - 1 + (j - 1) * k] = c[l - 1 + (j - 1) * k] + a[i - 1 + (j - 1) * m]
    for l in range(1, k + 1):
        if nc[l - 1] == 0:
            return ic1, nc, wss, 1
    for l in range(1, k + 1):
        aa = float(nc[l - 1])
        for j in range(1, n + 1):
            c[l - 1 + (j - 1) * k] = c[l - 1 + (j - 1) * k] / aa
        an2[l - 1] = aa / (aa + 1.0)
        an1[l - 1] = aa / (aa - 1.0) if 1.0 < aa else R8_HUGE
        itran[l - 1] = 1
        ncp[l - 1] = -1
    indx = [0]
    ifault = 2
    for _ in range(iters):
        _optra(a, m, n, c, k, ic1, ic2, nc, an1, an2, ncp, d, itran, live, indx)
        if indx[0] == m:
            ifault = 0
            break
        _qtran(a, m, n, c, k, ic1, ic2, nc, an1, an2, ncp, d, itran, indx)
        if k == 2:
            ifault = 0
            break
        for l in range(k):
            ncp[l] = 0
    for l in range(1, k + 1):
        wss[l - 1] = 0.0
        for j in range(1, n + 1):
            c[l - 1 + (j - 1) * k] = 0.0
    for i in range(1, m + 1):
        ii = ic1[i - 1]
        for j in range(1, n + 1):
            c[ii - 1 + (j - 1) * k] = c[ii - 1 + (j - 1) * k] + a[i - 1 + (j - 1) * m]
    for j in range(1, n + 1):
        for l in range(1, k + 1):
            c[l - 1 + (j - 1) * k] = c[l - 1 + (j - 1) * k] / float(nc[l - 1])
        for i in range(1, m + 1):
            ii = ic1[i - 1]
            da = a[i - 1 + (j - 1) * m] - c[ii - 1 + (j - 1) * k]
            wss[ii - 1] = wss[ii - 1] + da * da
    return ic1, nc, wss, ifault


def _dist2(a, m, n, c, k, i, l):
    s = 0.0
    for j in range(1, n + 1):
        df = a[i - 1 + (j - 1) * m] - c[l - 1 + (j - 1) * k]
        s = s + df * df
    return s


def _transfer(a, m, n, c, k, nc, an1, an2, ic1, ic2, i, l1, l2):
    al1 = float(nc[l1 - 1])
    alw = al1 - 1.0
    al2 = float(nc[l2 - 1])
    alt = al2 + 1.0
    for j in range(1, n + 1):
        c[l1 - 1 + (j - 1) * k] = (c[l1 - 1 + (j - 1) * k] * al1 - a[i - 1 + (j - 1) * m]) / alw
        c[l2 - 1 + (j - 1) * k] = (c[l2 - 1 + (j - 1) * k] * al2 + a[i - 1 + (j - 1) * m]) / alt
    nc[l1 - 1] -= 1
    nc[l2 - 1] += 1
    an2[l1 - 1] = alw / al1
    an1[l1 - 1] = alw / (alw - 1.0) if 1.0 < alw else R8_HUGE
    an1[l2 - 1] = alt / al2
    an2[l2 - 1] = alt / (alt + 1.0)
    ic1[i - 1] = l2
    ic2[i - 1] = l1


def _optra(a, m, n, c, k, ic1, ic2, nc, an1, an2, ncp, d, itran, live, indx):
    for l in range(1, k + 1):
        if itran[l - 1] == 1:
            live[l - 1] = m + 1
    for i in range(1, m + 1):
        indx[0] += 1
        l1, l2 = ic1[i - 1], ic2[i - 1]
        ll = l2
        if 1 < nc[l1 - 1]:
            if ncp[l1 - 1] != 0:
                d[i - 1] = _dist2(a, m, n, c, k, i, l1) * an1[l1 - 1]
            r2 = _dist2(a, m, n, c, k, i, l2) * an2[l2 - 1]
            for l in range(1, k + 1):
                if (i < live[l1 - 1] or i < live[l2 - 1]) and l != l1 and l != ll:
                    rr = r2 / an2[l - 1]
                    dc = _dist2(a, m, n, c, k, i, l)
                    if dc < rr:
                        r2 = dc * an2[l - 1]
                        l2 = l
            if d[i - 1] <= r2:
                ic2[i - 1] = l2
            else:
                indx[0] = 0
                live[l1 - 1] = m + i
                live[l2 - 1] = m + i
                ncp[l1 - 1] = i
                ncp[l2 - 1] = i
                _transfer(a, m, n, c, k, nc, an1, an2, ic1, ic2, i, l1, l2)
        if indx[0] == m:
            return
    for l in range(1, k + 1):
        itran[l - 1] = 0
        live[l - 1] = live[l - 1] - m


def _qtran(a, m, n, c, k, ic1, ic2, nc, an1, an2, ncp, d, itran, indx):
    icoun = istep = 0
    while True:
        for i in range(1, m + 1):
            icoun += 1
            istep += 1
            l1, l2 = ic1[i - 1], ic2[i - 1]
            if 1 < nc[l1 - 1]:
                if istep <= ncp[l1 - 1]:
                    d[i - 1] = _dist2(a, m, n, c, k, i, l1) * an1[l1 - 1]
                if istep < ncp[l1 - 1] or istep < ncp[l2 - 1]:
                    r2 = d[i - 1] / an2[l2 - 1]
                    dd = _dist2(a, m, n, c, k, i, l2)
                    if dd < r2:
                        icoun = 0
                        indx[0] = 0
                        itran[l1 - 1] = 1
                        itran[l2 - 1] = 1
                        ncp[l1 - 1] = istep + m
                        ncp[l2 - 1] = istep + m
                        _transfer(a, m, n, c, k, nc, an1, an2, ic1, ic2, i, l1, l2)
            if icoun == m:
                return


# ------------------------------------------------------------------------------------------
# MatePairEM (tools/MatePairEM.cpp)
# ------------------------------------------------------------------------------------------
class MatePairEM:
    def __init__(self, frag_mean, frag_sd, precision, min_cluster_size):   # Initialize :43-58
        self.mean, self.sd, self.min_size = frag_mean, frag_sd, float(min_cluster_size)
        x = -frag_sd * normal_01_cdf_inverse((1 - precision) / 2)
        self.min_prob = normalpdf(x, 0, frag_sd)
        self.max_frag = frag_mean + 3 * frag_sd
        self.kmeans_iter, self.lam, self.tol, self.kmax = 1000, 0.1, 0.001, 10

    @staticmethod
    def strand_remap(region, strand):   # :75-83
        return (region[0], region[1]) if strand == PLUS else (-region[1], -region[0])

    def pair_probability(self, x, y, u, a, b):   # :91-94
        return normalpdf(a + b - x - y, u, self.sd) * math.exp(-self.lam * max(0.0, x - a) - self.lam * max(0.0, y - b))

    def _exponents(self):
        N, K = self.N, self.K
        ex = [[0.0] * N for _ in range(K)]
        for i in range(N):
            for j in range(K):
                t = (self.A[j] + self.B[j] - self.X[i] - self.Y[i] - self.U[i]) / self.sd
                ex[j][i] = -0.5 * (t * t) - self.lam * max(0.0, self.X[i] - self.A[j]) - self.lam * max(0.0, self.Y[i] - self.B[j])
        return ex

    def log_likelihood(self):   # :96-137
        ex = self._exponents()
        LL = 0.0
        for i in range(self.N):
            maxexp = ex[0][i]
            for j in range(1, self.K):
                maxexp = max(maxexp, ex[j][i])
            s = 0.0
            for j in range(self.K):
                s += self.W[j] * math.exp(ex[j][i] - maxexp)
            if s == 0.0:
                return -DBL_MAX
            LL = LL + math.log(s) + maxexp
        return LL

    def update_responsibilities(self):   # :139-181
        ex = self._exponents()
        for i in range(self.N):
            ixo, iyo = self.ToXO[i], self.ToYO[i]
            maxexp = ex[0][i]
            for j in range(1, self.K):
                maxexp = max(maxexp, ex[j][i])
            norm = 0.0
            for j in range(self.K):
                norm += self.W[j] * math.exp(ex[j][i] - maxexp)
            if norm == 0.0:
                raise SystemExit("Error: norm != 0.0 failed")
            for j in range(self.K):
                self.R[j][i] = self.W[j] * math.exp(ex[j][i] - maxexp) / norm
                self.RXO[j][ixo] = self.R[j][i]
                self.RYO[j][iyo] = self.R[j][i]

    def max_likelihood(self, R, RXO, RYO):   # :192-325
        N, XO, YO = self.N, self.XO, self.YO
        SX, SY = [0.0] * N, [0.0] * N
        acc = 0.0
        for i in range(N):          # std::partial_sum: serial left-to-right
            acc = RXO[i] if i == 0 else acc + RXO[i]
            SX[i] = acc
        for i in range(N):
            acc = RYO[i] if i == 0 else acc + RYO[i]
            SY[i] = acc
        i = j = 0
        CX, CY, CS = [XO[0]], [YO[0]], [0.0]
        while i < N and j < N:
            if i + 1 < N and XO[i] == XO[i + 1]:
                i += 1
                continue
            if j + 1 < N and YO[j] == YO[j + 1]:
                j += 1
                continue
            if SX[i] == SY[j]:
                CX.append(XO[i]); CY.append(YO[j]); CS.append(SX[i])
                if i + 1 < N and j + 1 < N:
                    CX.append(XO[i + 1]); CY.append(YO[j + 1]); CS.append(SX[i])
                i += 1
                j += 1
            elif SX[i] < SY[j]:
                CX.append(XO[i]); CY.append(YO[j]); CS.append(SX[i])
                if i + 1 < N:
                    CX.append(XO[i + 1]); CY.append(YO[j]); CS.append(SX[i])
                i += 1
            else:
                CX.append(XO[i]); CY.append(YO[j]); CS.append(SY[j])
                if j + 1 < N:
                    CX.append(XO[i]); CY.append(YO[j + 1]); CS.append(SY[j])
                j += 1
        NK = 0.0
        for r in R:
            NK += r
        if NK == 0.0:
            return None
        RXYU = 0.0
        for t in range(N):
            RXYU += R[t] * (self.X[t] + self.Y[t] + self.U[t])
        var = self.sd * self.sd
        minindex = 0
        while minindex < len(CS):
            if (RXYU - NK * (CX[minindex] + CY[minindex])) / var + self.lam * CS[minindex] > 0:
                break
            minindex += 1
        if minindex >= len(CS):
            raise SystemExit("Error: MaxLikelihood ran past the breakpoints (undefined behaviour in the reference)")
        aplusb = (RXYU + var * self.lam * CS[minindex]) / NK
        if minindex == 0:
            min_a, max_a = CX[0], aplusb - CY[0]
            a = 0.5 * (min_a + max_a)
            b = aplusb - a
        elif CS[minindex] != CS[minindex - 1]:
            a, b = CX[minindex], CY[minindex]
        else:
            min_a = max(CX[minindex], aplusb - CY[minindex - 1])
            max_a = min(CX[minindex - 1], aplusb - CY[minindex])
            a = 0.5 * (min_a + max_a)
            b = aplusb - a
        return a, b

    def select_kkz(self, k):   # :327-386
        X, Y, N = self.X, self.Y, self.N
        l2max, imax = X[0] * Y[0], 0
        for i in range(1, N):
            l2 = X[i] * Y[i]
            if l2 > l2max:
                imax, l2max = i, l2
        A, B = [X[imax]], [Y[imax]]
        while len(A) < k:
            dist_min = [0.0] * N
            for i in range(N):
                md = (X[i] - A[0]) * (X[i] - A[0]) + (Y[i] - B[0]) * (Y[i] - B[0])
                for j in range(1, len(A)):
                    dj = (X[i] - A[j]) * (X[i] - A[j]) + (Y[i] - B[j]) * (Y[i] - B[j])
                    md = min(md, dj)
                dist_min[i] = md
            dmax, idx = dist_min[0], 0
            for i in range(N):
                if dist_min[i] > dmax:
                    dmax, idx = dist_min[i], i
            if dmax == 0.0:
                return None
            A.append(X[idx])
            B.append(Y[idx])
        return A, B

    def expectation_maximization(self):   # :388-494; returns log-likelihood or None
        N, K = self.N, self.K
        self.R = [[0.0] * N for _ in range(K)]
        self.RXO = [[0.0] * N for _ in range(K)]
        self.RYO = [[0.0] * N for _ in range(K)]
        self.W = (self.W + [0.0] * K)[:K]      # vector::resize keeps old values
        self.A = (self.A + [0.0] * K)[:K]
        self.B = (self.B + [0.0] * K)[:K]
        if K == 1 or K == N:
            for j in range(K):
                self.R[j] = [1.0 / K] * N
                self.RXO[j] = [1.0 / K] * N
                self.RYO[j] = [1.0 / K] * N
        else:
            seeds = self.select_kkz(K)
            if seeds is None:
                return None
            px, py = seeds
            a = list(self.Y) + list(self.X)       # both inserts are at begin(): [Y..., X...]
            c = list(py) + list(px)
            ic1, nc, wss, ifault = kmns(a, N, 2, c, K, self.kmeans_iter)
            if ifault == 1 or ifault == 3:
                raise SystemExit("Error: ifault != %d failed" % ifault)
            for i in range(N):
                for j in range(K):
                    v = 1.0 if j == ic1[i] - 1 else 0.0
                    self.R[j][i] = v
                    self.RXO[j][self.ToXO[i]] = v
                    self.RYO[j][self.ToYO[i]] = v
        last, valid = 0.0, False
        while True:
            for j in range(K):
                ab = self.max_likelihood(self.R[j], self.RXO[j], self.RYO[j])
                if ab is not None:
                    self.A[j], self.B[j] = ab
            for j in range(K):                        # UpdateMixWeights :183-190
                nk = 0.0
                for r in self.R[j]:
                    nk += r
                self.W[j] = nk / N
            ll = self.log_likelihood()
            if valid and abs(ll - last) < self.tol:
                break
            if valid and ll == -DBL_MAX:
                return None
            if valid and not (ll / last < 1.0000001):
                raise SystemExit("Error: likelihood check failed")
            last, valid = ll, True
            self.update_responsibilities()
        return last

    def do_clustering(self, mate_pairs):   # :540-636; mate_pairs: list of ((s1,e1),(s2,e2)) strand-remapped
        if len(mate_pairs) < self.min_size:
            return []
        N = self.N = len(mate_pairs)
        self.X = [float(mp[0][1]) for mp in mate_pairs]
        self.Y = [float(mp[1][1]) for mp in mate_pairs]
        self.U = [self.mean - (mp[0][1] - mp[0][0] + 1) - (mp[1][1] - mp[1][0] + 1) for mp in mate_pairs]
        ox = sorted(range(N), key=lambda i: (-self.X[i], i))
        self.XO = [self.X[i] for i in ox]
        self.ToXO = [0] * N
        for s, i in enumerate(ox):
            self.ToXO[i] = s
        oy = sorted(range(N), key=lambda i: (-self.Y[i], i))
        self.YO = [self.Y[i] for i in oy]
        self.ToYO = [0] * N
        for s, i in enumerate(oy):
            self.ToYO[i] = s
        # mW/mA/mB are members the reference never clears; a component with zero responsibility would
        # read stale values from an earlier bin pair.  Canonical here (and in the kernel): start at 0.
        self.W, self.A, self.B = [], [], []
        min_bic, k_min = None, 1
        for K in range(1, min(self.kmax, N) + 1):
            self.K = K
            ll = self.expectation_maximization()
            if ll is None:
                continue
            bic = -2.0 * ll + K * 2.0 * math.log(N)
            if min_bic is None or bic < min_bic:
                min_bic, k_min = bic, K
        self.K = k_min
        if self.expectation_maximization() is None:
            return []
        clusters = []
        for j in range(self.K):
            cl = [i for i in range(N) if self.pair_probability(self.X[i], self.Y[i], self.U[i], self.A[j], self.B[j]) > self.min_prob]
            if len(cl) >= self.min_size:
                clusters.append(cl)
        return clusters


# ------------------------------------------------------------------------------------------
# tools/clustermatepairs.cpp
# ------------------------------------------------------------------------------------------
BIN_LENGTH = 1 << 15


def get_bins(region, bin_length, extend):   # Binning::GetBins :152-162
    return list(range(cdiv(region[0] - extend, bin_length), cdiv(region[1] + extend, bin_length) + 1))


def pack_id(ref, strand, b):
    if ref >= (1 << 18):
        raise SystemExit("Packing failed, too many reference sequences")
    if b >= (1 << 13):
        raise SystemExit("Packing failed, chromosome too large")
    return ref + (strand << 18) + ((b & 0x1FFF) << 19)


def read_fragments(lines):
    """CompactAlignmentStream + FragmentAlignmentStream: yields lists of raw alignments per fragment run."""
    cur, cur_name = [], None
    for n, line in enumerate(lines, 1):
        line = line.rstrip("\n")
        if not line:
            raise SystemExit("Error: Empty alignment line %d" % n)
        f = line.split("\t")
        if len(f) < 6:
            raise SystemExit("Error: Format error for alignment line %d" % n)
        al = dict(fragment=f[0], readEnd=0 if f[1] == "1" else 1, reference=f[2], strand=MINUS if f[3] == "-" else PLUS,
                  region=(int(f[4]), int(f[5])))
        if cur and al["fragment"] != cur_name:
            yield cur
            cur = []
        cur_name = al["fragment"]
        cur.append(al)
    if cur:
        yield cur


def add_fragment(als, min_fusion_range, bin_pairs):
    """One fragment's alignments (dicts with frag, readEnd, ref, strand, region) into the map of bin pairs:
    CheckConcordant (:211-244), then AddBinPairs (:246-290)."""
    conc = [set(), set()]
    for a in als:
        for b in get_bins(a["region"], min_fusion_range, min_fusion_range):
            conc[a["readEnd"]].add((a["ref"], b))
    if conc[0] & conc[1]:
        return
    binned = [{}, {}]
    for a in als:
        for b in get_bins(a["region"], BIN_LENGTH, min_fusion_range):
            pid = pack_id(a["ref"], a["strand"], b)
            rs = a["region"][0] - b * BIN_LENGTH + BIN_LENGTH // 2
            re_ = a["region"][1] - b * BIN_LENGTH + BIN_LENGTH // 2
            if not (0 <= rs < 65536 and 0 <= re_ < 65536):
                raise SystemExit("Error: relativeStart out of range")
            binned[a["readEnd"]].setdefault(pid, []).append((a["frag"], a["readEnd"], rs, re_))
    for b1 in sorted(binned[0]):
        for b2 in sorted(binned[1]):
            if b1 < b2:
                e = bin_pairs.setdefault((b1, b2), ([], []))
                e[0].extend(binned[0][b1])
                e[1].extend(binned[1][b2])
            else:
                e = bin_pairs.setdefault((b2, b1), ([], []))
                e[0].extend(binned[1][b2])
                e[1].extend(binned[0][b1])


def clustermatepairs(lines, frag_mean, frag_sd, precision, min_cluster_size, em="python"):
    """Returns (output text, number of clusters).  em="c" runs MatePairEM::DoClustering through the C restatement
    (oracle/mpe_oracle.c, same arithmetic, ~100x faster) instead of the Python one below; tests compare the two."""
    min_fusion_range = int(frag_mean + 10 * frag_sd)
    ref_names, ref_index = [], {}
    bin_pairs = {}
    for raws in read_fragments(lines):
        als = []
        for r in raws:
            if r["reference"] not in ref_index:
                ref_index[r["reference"]] = len(ref_names)
                ref_names.append(r["reference"])
            als.append(dict(frag=int(r["fragment"]), readEnd=r["readEnd"], ref=ref_index[r["reference"]],
                            strand=r["strand"], region=r["region"]))
        add_fragment(als, min_fusion_range, bin_pairs)
    use_c = em == "c"
    em_obj = MatePairEM(frag_mean, frag_sd, precision, min_cluster_size)
    out = []
    cluster_id = 0
    for (b1, b2) in sorted(bin_pairs):
        p1, p2 = bin_pairs[(b1, b2)]
        if len(p1) < min_cluster_size or len(p2) < min_cluster_size:
            continue

        def unpack(pid, packed):
            ref, strand, b = pid & 0x3FFFF, (pid >> 18) & 1, pid >> 19
            return [dict(frag=f, readEnd=e, ref=ref, strand=strand,
                         region=(rs + b * BIN_LENGTH - BIN_LENGTH // 2, re_ + b * BIN_LENGTH - BIN_LENGTH // 2))
                    for (f, e, rs, re_) in packed]
        a1, a2 = unpack(b1, p1), unpack(b2, p2)
        fr1, fr2 = {}, {}
        for i, a in enumerate(a1):
            fr1.setdefault(a["frag"], []).append(i)
        for i, a in enumerate(a2):
            fr2.setdefault(a["frag"], []).append(i)
        fr2 = {k: v for k, v in fr2.items() if k in fr1}          # FilterUnmatched x2
        fr1 = {k: v for k, v in fr1.items() if k in fr2}

        def filter_overlapping(frs, als):                          # :316-358
            for k in frs:
                bins = [set(), set()]
                kept = []
                for idx in frs[k]:
                    a = als[idx]
                    rid = a["ref"] + (a["strand"] << 31)
                    rb = [(rid, b) for b in get_bins(a["region"], min_fusion_range, 0)]
                    if not any(x in bins[a["readEnd"]] for x in rb):
                        bins[a["readEnd"]].update(rb)
                        kept.append(idx)
                frs[k] = kept
        filter_overlapping(fr1, a1)
        filter_overlapping(fr2, a2)
        if len(fr1) < min_cluster_size or len(fr2) < min_cluster_size:
            continue
        pairs, seen = [], {}
        for k in sorted(fr1):                                      # GetAlignPairs :360-375 (canonical: ascending fragment)
            for i1 in fr1[k]:
                for i2 in fr2[k]:
                    if (i1, i2) not in seen:
                        seen[(i1, i2)] = len(pairs)
                        pairs.append((i1, i2))
        mps = [(MatePairEM.strand_remap(a1[i1]["region"], a1[i1]["strand"]),
                MatePairEM.strand_remap(a2[i2]["region"], a2[i2]["strand"])) for (i1, i2) in pairs]
        if use_c:
            from oracle import mpe_c
            found = mpe_c.do_clustering(frag_mean, frag_sd, em_obj.min_prob, min_cluster_size, mps)
        else:
            found = em_obj.do_clustering(mps)
        for cl in found:
            if len(cl) < min_cluster_size:
                continue
            used = set()
            for el in cl:
                i1, i2 = pairs[el]
                fidx = a1[i1]["frag"]
                if fidx in used:
                    continue
                used.add(fidx)
                for ce, a in ((0, a1[i1]), (1, a2[i2])):
                    out.append("%d\t%d\t%d\t%d\t%s\t%s\t%d\t%d\n" % (cluster_id, ce, a["frag"], a["readEnd"], ref_names[a["ref"]],
                                                                      "+" if a["strand"] == PLUS else "-", a["region"][0], a["region"][1]))
            cluster_id += 1
    return "".join(out), cluster_id
