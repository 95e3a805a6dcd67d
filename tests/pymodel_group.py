"""Independent Python big-integer model of ristretto255 (RFC 9496) on extended
Edwards25519 coordinates.  Test infrastructure: pins oracle/group.c and generates goldens."""
P = 2**255 - 19
D = (-121665 * pow(121666, -1, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)
Lq = 2**252 + 27742317777372353535851937790883648493


def _is_neg(x):
    return (x % P) & 1


def _abs(x):
    x %= P
    return P - x if x & 1 else x


def sqrt_ratio_m1(u, v):
    u %= P
    v %= P
    v3 = v * v % P * v % P
    v7 = v3 * v3 % P * v % P
    r = u * v3 % P * pow(u * v7 % P, (P - 5) // 8, P) % P
    check = v * r % P * r % P
    correct = check == u
    flipped = check == (-u) % P
    flipped_i = check == (-u) * SQRT_M1 % P
    if flipped or flipped_i:
        r = r * SQRT_M1 % P
    return (correct or flipped), _abs(r)


INVSQRT_A_MINUS_D = sqrt_ratio_m1(1, (-1 - D) % P)[1]
# RFC 9496 lists the odd root of a*d-1 (SURVEY.md A.2 sign trap)
_s = sqrt_ratio_m1((-D - 1) % P, 1)[1]
SQRT_AD_MINUS_ONE = _s if _s & 1 else P - _s
ONE_MINUS_D_SQ = (1 - D * D) % P
D_MINUS_ONE_SQ = (D - 1) ** 2 % P


class Pt:
    """extended coordinates"""

    def __init__(self, X, Y, Z, T):
        self.X, self.Y, self.Z, self.T = X % P, Y % P, Z % P, T % P

    @staticmethod
    def identity():
        return Pt(0, 1, 1, 0)

    def __add__(self, o):
        A = (self.Y - self.X) * (o.Y - o.X) % P
        B = (self.Y + self.X) * (o.Y + o.X) % P
        C = self.T * 2 * D % P * o.T % P
        Dd = self.Z * 2 * o.Z % P
        E, F, G, H = B - A, Dd - C, Dd + C, B + A
        return Pt(E * F, G * H, F * G, E * H)

    def __neg__(self):
        return Pt(-self.X, self.Y, self.Z, -self.T)

    def __sub__(self, o):
        return self + (-o)

    def __rmul__(self, k):
        k %= Lq
        acc, base = Pt.identity(), self
        while k:
            if k & 1:
                acc = acc + base
            base = base + base
            k >>= 1
        return acc

    def __eq__(self, o):
        return (self.X * o.Y - self.Y * o.X) % P == 0 or (self.Y * o.Y - self.X * o.X) % P == 0

    def encode(self):
        x0, y0, z0, t0 = self.X, self.Y, self.Z, self.T
        u1 = (z0 + y0) * (z0 - y0) % P
        u2 = x0 * y0 % P
        _, invsqrt = sqrt_ratio_m1(1, u1 * u2 % P * u2 % P)
        den1, den2 = invsqrt * u1 % P, invsqrt * u2 % P
        z_inv = den1 * den2 % P * t0 % P
        ix0, iy0 = x0 * SQRT_M1 % P, y0 * SQRT_M1 % P
        ench = den1 * INVSQRT_A_MINUS_D % P
        if _is_neg(t0 * z_inv):
            x, y, den_inv = iy0, ix0, ench
        else:
            x, y, den_inv = x0, y0, den2
        if _is_neg(x * z_inv):
            y = -y
        s = _abs(den_inv * (z0 - y))
        return s.to_bytes(32, "little")


def basepoint():
    y = 4 * pow(5, -1, P) % P
    x2 = (y * y - 1) * pow(D * y * y + 1, -1, P) % P
    x = pow(x2, (P + 3) // 8, P)
    if (x * x - x2) % P:
        x = x * SQRT_M1 % P
    if x & 1:
        x = P - x
    return Pt(x, y, 1, x * y)


def decode(b):
    s = int.from_bytes(b, "little")
    if s >= P or s & 1:
        return None
    ss = s * s % P
    u1, u2 = (1 - ss) % P, (1 + ss) % P
    u2s = u2 * u2 % P
    v = (-(D * u1 % P * u1) - u2s) % P
    ok, invsqrt = sqrt_ratio_m1(1, v * u2s % P)
    den_x = invsqrt * u2 % P
    den_y = invsqrt * den_x % P * v % P
    x = _abs(2 * s * den_x)
    y = u1 * den_y % P
    t = x * y % P
    if not ok or _is_neg(t) or y == 0:
        return None
    return Pt(x, y, 1, t)


def elligator(t):
    r = SQRT_M1 * t % P * t % P
    u = (r + 1) * ONE_MINUS_D_SQ % P
    v = (-1 - r * D) * (r + D) % P
    ok, s = sqrt_ratio_m1(u, v)
    sp = (-_abs(s * t)) % P
    if not ok:
        s, c = sp, r
    else:
        c = P - 1
    N = (c * (r - 1) % P * D_MINUS_ONE_SQ - v) % P
    w0 = 2 * s * v % P
    w1 = N * SQRT_AD_MINUS_ONE % P
    w2 = (1 - s * s) % P
    w3 = (1 + s * s) % P
    return Pt(w0 * w3, w2 * w1, w1 * w3, w0 * w2)


def from_uniform_bytes(b):
    t0 = int.from_bytes(b[:32], "little") & ((1 << 255) - 1)
    t1 = int.from_bytes(b[32:], "little") & ((1 << 255) - 1)
    return elligator(t0 % P) + elligator(t1 % P)
