"""Toom-Cook / Winograd minimal filtering F(m, r) in exact rational arithmetic - the derivation of the three matrices that
remora_amd/csrc/k_wino.hip (BT, AT) and engine.hip pack_conv (G) hold as constants for the fp32 5-tap stride-1 convolutions
(merge_conv1 / merge_conv2: models/ConvLSTM_w_ref.py:36-37,50, models/Conv_w_ref.py:35-38,54-55).

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py header for who may import this).  Nothing here is the reference's algorithm: the
reference computes these layers with torch.nn.Conv1d; F(m, r) is another exact evaluation of the same sums
y[i] = sum_k g[k] d[i + k], and tests/test_host_cpu.py holds the shipped constants to this derivation.

With evaluation matrices E_k (row for point a: 1, a, a^2 .. a^(k-1); row for infinity: 0 .. 0 1) and n = m + r - 1 points, the
product polynomial of degrees m - 1 and r - 1 is interpolated by C^-1 with C = E_n; transposing that linear-convolution
algorithm gives the correlation form   y = E_m^T [ (E_r g) * (C^-T d) ]   =  AT [ (G g) * (BT d) ].   Rows of BT are scaled to
coprime integers and the scale moved into G (a diagonal rescaling between the two leaves the product unchanged)."""
from fractions import Fraction as Fr
from math import gcd, lcm

F45_POINTS = (0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2), None)  # None = the point at infinity
F45_KERNEL_ORDER = (1, 2, 3, 4, 0, 5, 6, 7)                # x order of k_wino.hip: wave half 0 (+-1, +-2), half 1 (0, +-1/2, inf)


def _eval_matrix(points, k):
    return [[(Fr(p) ** j) if p is not None else Fr(int(j == k - 1)) for j in range(k)] for p in points]


def _inverse(M):
    n = len(M)
    A = [row[:] + [Fr(int(i == j)) for j in range(n)] for i, row in enumerate(M)]
    for c in range(n):
        piv = next(i for i in range(c, n) if A[i][c] != 0)
        A[c], A[piv] = A[piv], A[c]
        pv = A[c][c]
        A[c] = [x / pv for x in A[c]]
        for i in range(n):
            if i != c and A[i][c] != 0:
                f = A[i][c]
                A[i] = [x - f * y for x, y in zip(A[i], A[c])]
    return [row[n:] for row in A]


def matrices(m, r, points):
    """(AT [m x n], G [n x r], BT [n x n]) as lists of Fractions for F(m, r) at `points` (n = m + r - 1 of them)."""
    n = m + r - 1
    assert len(points) == n
    cinv = _inverse(_eval_matrix(points, n))
    BT = [[cinv[j][i] for j in range(n)] for i in range(n)]
    G = _eval_matrix(points, r)
    Em = _eval_matrix(points, m)
    AT = [[Em[i][j] for i in range(n)] for j in range(m)]
    for i in range(n):  # integer rows of BT, the scale into G
        den = 1
        for x in BT[i]:
            den = lcm(den, x.denominator)
        g = 0
        for x in BT[i]:
            g = gcd(g, abs(int(x * den)))
        s = Fr(den, g)
        BT[i] = [x * s for x in BT[i]]
        G[i] = [x / s for x in G[i]]
    return AT, G, BT


def correlate(g, d, m):
    return [sum(g[k] * d[i + k] for k in range(len(g))) for i in range(m)]


def apply(AT, G, BT, g, d):
    """y = AT [(G g) * (BT d)] with whatever number type g and d carry."""
    u = [sum(G[x][k] * g[k] for k in range(len(g))) for x in range(len(G))]
    v = [sum(BT[x][j] * d[j] for j in range(len(d))) for x in range(len(BT))]
    return [sum(AT[i][x] * u[x] * v[x] for x in range(len(u))) for i in range(len(AT))]
