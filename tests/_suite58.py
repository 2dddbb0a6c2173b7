"""The problem families of the reference's benchmark module (benchmarks/lsq_problems.py:1003-1018:
58 problems = 32 unbounded + 26 bounded start-point / box variants of 33 families), restated.

Nothing here is copied from the reference: every family is written down from its mathematical
definition (Moré, Garbow, Hillstrom, "Testing unconstrained optimization software", ACM TOMS 7,
1981 — numbering below; the data-fitting families are the classical Osborne / Meyer / Kowalik-
Osborne data sets of the same paper).  What makes them THE reference's problems is data:
tests/golden/suite58.json holds, per problem, the start point and box the reference's factory
uses (captured by tests/golden/make_golden.py, which also checks every family here against the
reference's own residuals and Jacobians at random points to 1e-11), and what the reference's
public drivers returned on them.

One family is defined by DATA rather than by a formula alone: CoatingThickness (MINPACK-2 "coating
thickness standardization", n = 134, m = 252) fits two bilinear models to 63 measurements whose 252-entry
table (abscissae xi [2][63], ordinates y [126], two weights) is captured by make_golden.py into the
fixture like the start points are (`problems[i]["data"]`); `coating_thickness(data)` restates the model.

FAMILIES: reference factory name -> callable(n_or_m_hint) -> (fun, jac)
DATA_FAMILIES: ... -> callable(data) -> (fun, jac);  `functions(problem_record)` serves both.
"""
import numpy as np


def rosenbrock():                                             # MGH 1
    def fun(x):
        return np.array([10.0 * (x[1] - x[0] ** 2), 1.0 - x[0]])

    def jac(x):
        return np.array([[-20.0 * x[0], 10.0], [-1.0, 0.0]])
    return fun, jac


def freudenstein_roth():                                      # MGH 2
    def fun(x):
        a, b = x
        return np.array([a - 13.0 + ((5.0 - b) * b - 2.0) * b, a - 29.0 + ((b + 1.0) * b - 14.0) * b])

    def jac(x):
        b = x[1]
        return np.array([[1.0, -3.0 * b * b + 10.0 * b - 2.0], [1.0, 3.0 * b * b + 2.0 * b - 14.0]])
    return fun, jac


def powell_badly_scaled():                                    # MGH 3
    def fun(x):
        return np.array([1.0e4 * x[0] * x[1] - 1.0, np.exp(-x[0]) + np.exp(-x[1]) - 1.0001])

    def jac(x):
        return np.array([[1.0e4 * x[1], 1.0e4 * x[0]], [-np.exp(-x[0]), -np.exp(-x[1])]])
    return fun, jac


def brown_badly_scaled():                                     # MGH 4
    def fun(x):
        return np.array([x[0] - 1.0e6, x[1] - 2.0e-6, x[0] * x[1] - 2.0])

    def jac(x):
        return np.array([[1.0, 0.0], [0.0, 1.0], [x[1], x[0]]])
    return fun, jac


def beale():                                                  # MGH 5
    y = np.array([1.5, 2.25, 2.625])
    k = np.array([1.0, 2.0, 3.0])

    def fun(x):
        return y - x[0] * (1.0 - x[1] ** k)

    def jac(x):
        J = np.empty((3, 2))
        J[:, 0] = -(1.0 - x[1] ** k)
        J[:, 1] = x[0] * k * x[1] ** (k - 1.0)
        return J
    return fun, jac


def jennrich_sampson(m=10):                                   # MGH 6
    i = np.arange(1, m + 1, dtype=float)

    def fun(x):                       # (the reference poses it with the opposite sign to MGH)
        return np.exp(i * x[0]) + np.exp(i * x[1]) - (2.0 + 2.0 * i)

    def jac(x):
        return np.column_stack([i * np.exp(i * x[0]), i * np.exp(i * x[1])])
    return fun, jac


def helical_valley():                                         # MGH 7
    def angle(x):
        t = np.arctan(x[1] / x[0]) / (2.0 * np.pi)
        return t + 0.5 if x[0] < 0 else t

    def fun(x):
        return np.array([10.0 * (x[2] - 10.0 * angle(x)), 10.0 * (np.hypot(x[0], x[1]) - 1.0), x[2]])

    def jac(x):
        r2 = x[0] ** 2 + x[1] ** 2
        r = np.sqrt(r2)
        c = 50.0 / (np.pi * r2)
        return np.array([[c * x[1], -c * x[0], 10.0],
                         [10.0 * x[0] / r, 10.0 * x[1] / r, 0.0],
                         [0.0, 0.0, 1.0]])
    return fun, jac


def gaussian():                                               # MGH 9
    y = np.array([0.0009, 0.0044, 0.0175, 0.0540, 0.1295, 0.2420, 0.3521, 0.3989,
                  0.3521, 0.2420, 0.1295, 0.0540, 0.0175, 0.0044, 0.0009])
    t = (8.0 - np.arange(1, 16)) / 2.0

    def fun(x):
        return x[0] * np.exp(-0.5 * x[1] * (t - x[2]) ** 2) - y

    def jac(x):
        d = t - x[2]
        e = np.exp(-0.5 * x[1] * d ** 2)
        return np.column_stack([e, -0.5 * x[0] * d ** 2 * e, x[0] * x[1] * d * e])
    return fun, jac


def meyer():                                                  # MGH 10 (thermistor resistance)
    y = np.array([34780.0, 28610.0, 23650.0, 19630.0, 16370.0, 13720.0, 11540.0, 9744.0,
                  8261.0, 7030.0, 6005.0, 5147.0, 4427.0, 3820.0, 3307.0, 2872.0])
    t = 5.0 + 45.0 * np.arange(1, 17)     # (the reference's abscissae; MGH has 45 + 5 i)

    def fun(x):
        return x[0] * np.exp(x[1] / (t + x[2])) - y

    def jac(x):
        e = np.exp(x[1] / (t + x[2]))
        return np.column_stack([e, x[0] * e / (t + x[2]), -x[0] * x[1] * e / (t + x[2]) ** 2])
    return fun, jac


def gulf(m=100):                                              # MGH 11
    t = np.arange(1, m + 1) / 100.0
    y = 25.0 + (-50.0 * np.log(t)) ** (2.0 / 3.0)

    def fun(x):
        return np.exp(-np.abs(y - x[1]) ** x[2] / x[0]) - t

    def jac(x):
        d = np.abs(y - x[1])
        p = d ** x[2]
        e = np.exp(-p / x[0])
        # d|y - b|^c / db = -c |y - b|^(c-1) sign(y - b)
        return np.column_stack([e * p / x[0] ** 2,
                                e * x[2] * d ** (x[2] - 1.0) * np.sign(y - x[1]) / x[0],
                                -e * p * np.log(d) / x[0]])
    return fun, jac


def box3d(m=10):                                              # MGH 12
    t = 0.1 * np.arange(1, m + 1)
    c = np.exp(-t) - np.exp(-10.0 * t)

    def fun(x):
        return np.exp(-t * x[0]) - np.exp(-t * x[1]) - x[2] * c

    def jac(x):
        return np.column_stack([-t * np.exp(-t * x[0]), t * np.exp(-t * x[1]), -c])
    return fun, jac


def powell_singular():                                        # MGH 13
    r5, r10 = np.sqrt(5.0), np.sqrt(10.0)

    def fun(x):
        return np.array([x[0] + 10.0 * x[1], r5 * (x[2] - x[3]), (x[1] - 2.0 * x[2]) ** 2,
                         r10 * (x[0] - x[3]) ** 2])

    def jac(x):
        a, b = x[1] - 2.0 * x[2], x[0] - x[3]
        return np.array([[1.0, 10.0, 0.0, 0.0], [0.0, 0.0, r5, -r5],
                         [0.0, 2.0 * a, -4.0 * a, 0.0], [2.0 * r10 * b, 0.0, 0.0, -2.0 * r10 * b]])
    return fun, jac


def wood():                                                   # MGH 14
    r90, r10 = np.sqrt(90.0), np.sqrt(10.0)

    def fun(x):
        return np.array([10.0 * (x[1] - x[0] ** 2), 1.0 - x[0], r90 * (x[3] - x[2] ** 2), 1.0 - x[2],
                         r10 * (x[1] + x[3] - 2.0), (x[1] - x[3]) / r10])

    def jac(x):
        return np.array([[-20.0 * x[0], 10.0, 0.0, 0.0], [-1.0, 0.0, 0.0, 0.0],
                         [0.0, 0.0, -2.0 * r90 * x[2], r90], [0.0, 0.0, -1.0, 0.0],
                         [0.0, r10, 0.0, r10], [0.0, 1.0 / r10, 0.0, -1.0 / r10]])
    return fun, jac


def kowalik_osborne():                                        # MGH 15 (enzyme reaction)
    y = np.array([0.1957, 0.1947, 0.1735, 0.1600, 0.0844, 0.0627, 0.0456, 0.0342, 0.0323, 0.0235, 0.0246])
    u = np.array([4.0, 2.0, 1.0, 0.5, 0.25, 0.167, 0.125, 0.1, 0.0833, 0.0714, 0.0625])

    def fun(x):                       # model - data, as the reference poses it
        return x[0] * (u * u + u * x[1]) / (u * u + u * x[2] + x[3]) - y

    def jac(x):
        num = u * u + u * x[1]
        den = u * u + u * x[2] + x[3]
        return np.column_stack([num / den, x[0] * u / den, -x[0] * num * u / den ** 2,
                                -x[0] * num / den ** 2])
    return fun, jac


def brown_dennis(m=20):                                       # MGH 16
    t = np.arange(1, m + 1) / 5.0

    def fun(x):
        return (x[0] + t * x[1] - np.exp(t)) ** 2 + (x[2] + x[3] * np.sin(t) - np.cos(t)) ** 2

    def jac(x):
        a = x[0] + t * x[1] - np.exp(t)
        b = x[2] + x[3] * np.sin(t) - np.cos(t)
        return np.column_stack([2.0 * a, 2.0 * a * t, 2.0 * b, 2.0 * b * np.sin(t)])
    return fun, jac


def osborne1():                                               # MGH 17 (exponential fitting)
    y = np.array([0.844, 0.908, 0.932, 0.936, 0.925, 0.908, 0.881, 0.850, 0.818, 0.784, 0.751,
                  0.718, 0.685, 0.658, 0.628, 0.603, 0.580, 0.558, 0.538, 0.522, 0.506, 0.490,
                  0.478, 0.467, 0.457, 0.448, 0.438, 0.431, 0.424, 0.420, 0.414, 0.411, 0.406])
    t = 10.0 * np.arange(33)

    def fun(x):                       # model - data, as the reference poses it
        return x[0] + x[1] * np.exp(-t * x[3]) + x[2] * np.exp(-t * x[4]) - y

    def jac(x):
        e4, e5 = np.exp(-t * x[3]), np.exp(-t * x[4])
        return np.column_stack([np.ones(33), e4, e5, -x[1] * t * e4, -x[2] * t * e5])
    return fun, jac


def biggs_exp6(m=13):                                         # MGH 18
    t = 0.1 * np.arange(1, m + 1)
    y = np.exp(-t) - 5.0 * np.exp(-10.0 * t) + 3.0 * np.exp(-4.0 * t)

    def fun(x):
        return x[2] * np.exp(-t * x[0]) - x[3] * np.exp(-t * x[1]) + x[5] * np.exp(-t * x[4]) - y

    def jac(x):
        e1, e2, e5 = np.exp(-t * x[0]), np.exp(-t * x[1]), np.exp(-t * x[4])
        return np.column_stack([-t * x[2] * e1, t * x[3] * e2, e1, -e2, -t * x[5] * e5, e5])
    return fun, jac


def osborne2():                                               # MGH 19 (Gaussian fitting I)
    y = np.array([1.366, 1.191, 1.112, 1.013, 0.991, 0.885, 0.831, 0.847, 0.786, 0.725, 0.746,
                  0.679, 0.608, 0.655, 0.616, 0.606, 0.602, 0.626, 0.651, 0.724, 0.649, 0.649,
                  0.694, 0.644, 0.624, 0.661, 0.612, 0.558, 0.533, 0.495, 0.500, 0.423, 0.395,
                  0.375, 0.372, 0.391, 0.396, 0.405, 0.428, 0.429, 0.523, 0.562, 0.607, 0.653,
                  0.672, 0.708, 0.633, 0.668, 0.645, 0.632, 0.591, 0.559, 0.597, 0.625, 0.739,
                  0.710, 0.729, 0.720, 0.636, 0.581, 0.428, 0.292, 0.162, 0.098, 0.054])
    t = np.arange(65) / 10.0

    def parts(x):
        return (np.exp(-t * x[4]), np.exp(-(t - x[8]) ** 2 * x[5]), np.exp(-(t - x[9]) ** 2 * x[6]),
                np.exp(-(t - x[10]) ** 2 * x[7]))

    def fun(x):
        e0, e1, e2, e3 = parts(x)
        return x[0] * e0 + x[1] * e1 + x[2] * e2 + x[3] * e3 - y     # model - data

    def jac(x):
        e0, e1, e2, e3 = parts(x)
        J = np.empty((65, 11))
        J[:, 0], J[:, 1], J[:, 2], J[:, 3] = e0, e1, e2, e3
        J[:, 4] = -x[0] * t * e0
        J[:, 5] = -x[1] * (t - x[8]) ** 2 * e1
        J[:, 6] = -x[2] * (t - x[9]) ** 2 * e2
        J[:, 7] = -x[3] * (t - x[10]) ** 2 * e3
        J[:, 8] = 2.0 * x[1] * x[5] * (t - x[8]) * e1
        J[:, 9] = 2.0 * x[2] * x[6] * (t - x[9]) * e2
        J[:, 10] = 2.0 * x[3] * x[7] * (t - x[10]) * e3
        return J
    return fun, jac


def watson(n):                                                # MGH 20 (m = 31)
    t = np.arange(1, 30) / 29.0
    j = np.arange(n)

    def fun(x):
        f = np.empty(31)
        P = t[:, None] ** j                                    # t_i^(j-1), j = 1..n
        s = P.dot(x)
        d = (P[:, :n - 1] * j[1:]).dot(x[1:])                  # sum (j-1) x_j t^(j-2)
        f[:29] = d - s * s - 1.0
        f[29] = x[0]
        f[30] = x[1] - x[0] ** 2 - 1.0
        return f

    def jac(x):
        J = np.zeros((31, n))
        P = t[:, None] ** j
        s = P.dot(x)
        D = np.zeros((29, n))
        D[:, 1:] = P[:, :n - 1] * j[1:]
        J[:29] = D - 2.0 * s[:, None] * P
        J[29, 0] = 1.0
        J[30, 0] = -2.0 * x[0]
        J[30, 1] = 1.0
        return J
    return fun, jac


def penalty1(n=10):                                           # MGH 23
    ra = np.sqrt(1.0e-5)

    def fun(x):
        return np.append(ra * (x - 1.0), x.dot(x) - 0.25)

    def jac(x):
        return np.vstack([ra * np.eye(n), 2.0 * x])
    return fun, jac


def penalty2(n):                                              # MGH 24 (m = 2 n)
    ra = np.sqrt(1.0e-5)
    i = np.arange(2, n + 1)
    y = np.exp(i / 10.0) + np.exp((i - 1) / 10.0)
    w = n - np.arange(1, n + 1) + 1.0

    def fun(x):
        f = np.empty(2 * n)
        f[0] = x[0]                    # (the reference's variant; MGH has x_1 - 0.2)
        f[1:n] = ra * (np.exp(x[1:] / 10.0) + np.exp(x[:-1] / 10.0) - y)
        f[n:2 * n - 1] = ra * (np.exp(x[1:] / 10.0) - np.exp(-0.1))
        f[2 * n - 1] = w.dot(x * x) - 1.0
        return f

    def jac(x):
        J = np.zeros((2 * n, n))
        J[0, 0] = 1.0
        e = ra * np.exp(x / 10.0) / 10.0
        r = np.arange(1, n)
        J[r, r] = e[1:]
        J[r, r - 1] = e[:-1]
        J[n - 1 + r, r] = e[1:]
        J[2 * n - 1] = 2.0 * w * x
        return J
    return fun, jac


def trigonometric(n=10):                                      # MGH 26
    i = np.arange(1, n + 1, dtype=float)

    def fun(x):
        return n - np.cos(x).sum() + i * (1.0 - np.cos(x)) - np.sin(x)

    def jac(x):
        J = np.tile(np.sin(x), (n, 1))
        J[np.arange(n), np.arange(n)] += i * np.sin(x) - np.cos(x)
        return J
    return fun, jac


def chebyquad(n):                                             # MGH 35 (m = n)
    k = np.arange(1, n + 1)
    integral = np.where(k % 2 == 0, -1.0 / np.where(k % 2 == 0, k * k - 1.0, 1.0), 0.0)

    def polys(x):
        """T_k(2 x - 1) and derivatives w.r.t. x, k = 0..n, rows by degree."""
        z = 2.0 * x - 1.0
        T = np.empty((n + 1, x.size))
        dT = np.empty((n + 1, x.size))
        T[0], dT[0] = 1.0, 0.0
        T[1], dT[1] = z, 2.0
        for d in range(2, n + 1):
            T[d] = 2.0 * z * T[d - 1] - T[d - 2]
            dT[d] = 4.0 * T[d - 1] + 2.0 * z * dT[d - 1] - dT[d - 2]
        return T, dT

    def fun(x):
        T, _ = polys(x)
        return T[1:].mean(axis=1) - integral

    def jac(x):
        _, dT = polys(x)
        return dT[1:] / n
    return fun, jac


# reference factory name (benchmarks/lsq_problems.py) -> restated family
FAMILIES = {
    "Rosenbrock": rosenbrock, "FreudensteinAndRoth": freudenstein_roth,
    "PowellBadlyScaled": powell_badly_scaled, "BrownBadlyScaled": brown_badly_scaled,
    "Beale": beale, "JenrichAndSampson10": lambda: jennrich_sampson(10),
    "HelicalValley": helical_valley, "GaussianFittingII": gaussian,
    "ThermistorResistance": meyer, "GulfRnD": lambda: gulf(100), "Box3D": lambda: box3d(10),
    "ExtendedPowellSingular": powell_singular, "Wood": wood, "EnzymeReaction": kowalik_osborne,
    "BrownAndDennis": lambda: brown_dennis(20), "ExponentialFitting": osborne1,
    "Biggs": lambda: biggs_exp6(13), "GaussianFittingI": osborne2,
    "Watson6": lambda: watson(6), "Watson9": lambda: watson(9), "Watson12": lambda: watson(12),
    "Watson20": lambda: watson(20), "PenaltyI": lambda: penalty1(10),
    "PenaltyII4": lambda: penalty2(4), "PenaltyII10": lambda: penalty2(10),
    "Trigonometric": lambda: trigonometric(10),
    "ChebyshevQuadrature7": lambda: chebyquad(7), "ChebyshevQuadrature8": lambda: chebyquad(8),
    "ChebyshevQuadrature9": lambda: chebyquad(9), "ChebyshevQuadrature10": lambda: chebyquad(10),
    "ChebyshevQuadrature11": lambda: chebyquad(11),
}


def coating_thickness(data):                                  # MINPACK-2 (Averick, Carter, More', Xue 1992)
    """Errors-in-variables fit of two bilinear response surfaces
        z_k(u, v) = a_k + b_k u + c_k v + d_k u v,   k = 1, 2,
    to q = 63 measurements: the abscissae (u_i, v_i) = table + their own corrections (the last 2 q
    unknowns), x[0:4] = (a, b, c, d)_1, x[4:8] = (a, b, c, d)_2.  Residuals: z_1 - y_1, z_2 - y_2 and the
    weighted corrections w_1 du, w_2 dv (m = 4 q)."""
    tab = np.array([[float.fromhex(v) for v in row] for row in data["xi"]])
    y = np.array([float.fromhex(v) for v in data["y"]])
    w1, w2 = float.fromhex(data["scale1"]), float.fromhex(data["scale2"])
    q = tab.shape[1]

    def surfaces(x):
        u = tab[0] + x[8:8 + q]
        v = tab[1] + x[8 + q:8 + 2 * q]
        return u, v

    def fun(x):
        x = np.asarray(x, float)
        u, v = surfaces(x)
        out = np.empty(4 * q)
        for k in range(2):
            a, b, c, d = x[4 * k:4 * k + 4]
            out[k * q:(k + 1) * q] = a + b * u + c * v + d * u * v - y[k * q:(k + 1) * q]
        out[2 * q:3 * q] = w1 * x[8:8 + q]
        out[3 * q:] = w2 * x[8 + q:8 + 2 * q]
        return out

    def jac(x):
        x = np.asarray(x, float)
        u, v = surfaces(x)
        J = np.zeros((4 * q, 8 + 2 * q))
        i = np.arange(q)
        for k in range(2):
            a, b, c, d = x[4 * k:4 * k + 4]
            rows = k * q + i
            J[rows, 4 * k] = 1.0
            J[rows, 4 * k + 1] = u
            J[rows, 4 * k + 2] = v
            J[rows, 4 * k + 3] = u * v
            J[rows, 8 + i] = b + d * v                        # d z_k / d u_i
            J[rows, 8 + q + i] = c + d * u                    # d z_k / d v_i
        J[2 * q + i, 8 + i] = w1
        J[3 * q + i, 8 + q + i] = w2
        return J
    return fun, jac


DATA_FAMILIES = {"CoatingThickness": coating_thickness}
NOT_RESTATED = {}


def functions(problem):
    """(fun, jac) of a problem record of tests/golden/suite58.json"""
    fam = problem["family"]
    if fam in DATA_FAMILIES:
        return DATA_FAMILIES[fam](problem["data"])
    return FAMILIES[fam]()


def family_of(problem_name):
    """'Watson9_B' / 'Rosenbrock_B_3' / 'Beale' -> factory name."""
    base = problem_name.split("_B")[0]
    return base
