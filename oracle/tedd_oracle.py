"""CPU oracle for the time-stepping hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This module is a NumPy/SciPy restatement of the reference algorithm
(`src/prob1base.py`, `src/prob1_mms_cases.py`, `src/mms_trial_utils.py` of
phao/NA-nonlinear-temperature-enhanced-diffusion-model-DD).  It exists to check the
HIP path and to serve as the timed CPU baseline.  Only `tests/`, `bench.py`'s
`cpu_baseline` leg and `__graft_entry__.smoke()` may import it; the product package
never does.

Parity status: PINNED.  `tests/test_oracle_vs_golden.py` checks every function here
against vectors produced by running the reference itself (`tests/golden/*.npz`,
written by `tests/golden/make_golden.py`) and against the error norms printed in the
reference's executed notebooks (`tests/golden/published_notebook_errors.json`).

Conventions (reference `src/prob1base.py:243-248,287-315,353`): arrays are float64,
C-order, shape (N+1, M+1); index i runs along x (slow axis), j along y (fast axis).
`h[i] = x[i]-x[i-1]` with `h[0] = inf`; `hm[i] = (h[i]+h[i+1])/2` for 1<=i<=N-1 and
`hm[0] = hm[N] = inf` (likewise k, km along y) so boundary quotients vanish.

Third-party arithmetic on the path: `scipy.sparse.linalg.spsolve` (SuperLU; the
reference pins scipy 1.15.2) solves the 5-diagonal Newton systems exactly, as the
reference does at `src/prob1base.py:2103,2130`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

CONST_NAMES = ("R0", "Ea", "K1", "K2", "K3", "K4", "DT", "Dl_max", "phi_l", "gamma_T",
               "Kd", "Sd", "Dd_max", "phi_d", "phi_T", "r_sp", "T_ref")
VARS = ("cp", "T", "cl", "cd", "cs")


# --------------------------------------------------------------------------------------
# model laws                                      reference: src/prob1base.py:28-217
# --------------------------------------------------------------------------------------
class Model:
    """Constitutive laws Dl, V1, V2, Dd.  kind=1: DefaultModel01, kind=2: DefaultModel02
    (Dd evaluated at T + T_ref, `src/prob1base.py:209-217`)."""

    def __init__(self, consts, kind=2):
        if isinstance(consts, dict):
            consts = [consts[n] for n in CONST_NAMES]
        consts = [float(c) for c in consts]
        assert len(consts) == len(CONST_NAMES)
        for n, c in zip(CONST_NAMES, consts):
            setattr(self, n, c)
        self.kind = int(kind)

    def consts(self):
        return np.array([getattr(self, n) for n in CONST_NAMES])

    def Dl(self, c):  # :96-112
        return self.Dl_max * np.exp(-self.phi_l * c)

    def dDl(self, c):
        return (-self.phi_l) * self.Dl(c)

    def V1(self, T):  # :114-131  (V2 == 0, :133-138)
        return self.gamma_T * T

    def _Teff(self, T):
        return T + self.T_ref if self.kind == 2 else T

    def Dd(self, c, T):  # :140-202 ; zero where the effective temperature is exactly 0
        c = np.asarray(c, dtype=float)
        Te = np.asarray(self._Teff(np.asarray(T, dtype=float)))
        out = np.zeros_like(Te)
        nz = Te != 0
        out[nz] = self.Dd_max * np.exp(-self.phi_d * c[nz]) * np.exp(-self.phi_T / Te[nz])
        return out

    def dDd_dc(self, c, T):
        return -self.phi_d * self.Dd(c, T)

    def dDd_dT(self, c, T):
        Te = np.asarray(self._Teff(np.asarray(T, dtype=float)))
        out = self.Dd(c, T)
        nz = Te != 0
        out[nz] *= self.phi_T / (Te[nz] ** 2)
        return out


def heaviside_reg(x, eta):  # src/prob1base.py:3452-3466
    with np.errstate(over="ignore"):
        return 1.0 / (1.0 + np.exp(-eta * x))


def reaction_gain(cs, react, eta):
    """g(cs) in R = Kd g(cs) (1+cl) (Sd-cd): H_eta for 'reg' (:3592-3593), identity for
    'cs3' (:2875-2876), the sharp Heaviside (cs > 0) for 'hcs' (:3336-3337)."""
    if react == "reg":
        return heaviside_reg(cs, eta)
    if react == "hcs":
        return (cs > 0).astype(float)
    if react == "cs3":
        return cs
    raise ValueError(react)


# --------------------------------------------------------------------------------------
# mesh + discrete operators                       reference: src/prob1base.py:220-490, 1499-1550
# --------------------------------------------------------------------------------------
class Mesh:
    def __init__(self, x, y):
        x = np.asarray(x, dtype=float)
        y = np.asarray(y, dtype=float)
        self.x, self.y = x, y
        self.N, self.M = len(x) - 1, len(y) - 1
        inf = np.array([np.inf])
        self.h = np.concatenate([inf, np.diff(x)])
        self.k = np.concatenate([inf, np.diff(y)])
        self.hm = np.concatenate([(self.h[:-1] + self.h[1:]) * 0.5, inf])
        self.km = np.concatenate([(self.k[:-1] + self.k[1:]) * 0.5, inf])
        self.X, self.Y = np.meshgrid(x, y, indexing="ij")
        self.shape = (self.N + 1, self.M + 1)
        self.mask = np.zeros(self.shape)
        self.mask[1:-1, 1:-1] = 1.0
        # column/row broadcast helpers
        self.hcol = self.h[:, None]
        self.hmcol = self.hm[:, None]
        self.krow = self.k[None, :]
        self.kmrow = self.km[None, :]

    @staticmethod
    def uniform(N, M=None):
        M = N if M is None else M
        return Mesh(np.linspace(0, 1, N + 1), np.linspace(0, 1, M + 1))

    def zeros(self):
        return np.zeros(self.shape)

    # face means / backward differences live on index i (face between i-1 and i)
    def mean_x(self, u):
        out = np.zeros_like(u)
        out[1:, :] = 0.5 * (u[1:, :] + u[:-1, :])
        return out

    def mean_y(self, u):
        out = np.zeros_like(u)
        out[:, 1:] = 0.5 * (u[:, 1:] + u[:, :-1])
        return out

    def dminus_x(self, u):
        out = np.zeros_like(u)
        out[1:, :] = (u[1:, :] - u[:-1, :]) / self.hcol[1:, :]
        return out

    def dminus_y(self, u):
        out = np.zeros_like(u)
        out[:, 1:] = (u[:, 1:] - u[:, :-1]) / self.krow[:, 1:]
        return out

    # forward differences of face quantities, interior rows/cols only
    def dstar_x(self, q):
        out = np.zeros_like(q)
        out[1:-1, :] = (q[2:, :] - q[1:-1, :]) / self.hmcol[1:-1, :]
        return out

    def dstar_y(self, q):
        out = np.zeros_like(q)
        out[:, 1:-1] = (q[:, 2:] - q[:, 1:-1]) / self.kmrow[:, 1:-1]
        return out

    # discrete norms, src/prob1base.py:387-433
    def normH_sq(self, u):
        w = self.hmcol[1:-1, :] * self.kmrow[:, 1:-1]
        return float(np.sum(u[1:-1, 1:-1] ** 2 * w))

    def normp_sq(self, ux, uy):
        a = np.sum(ux[1:, 1:-1] ** 2 * (self.hcol[1:, :] * self.kmrow[:, 1:-1]))
        b = np.sum(uy[1:-1, 1:] ** 2 * (self.hmcol[1:-1, :] * self.krow[:, 1:]))
        return float(a + b)

    def grad_err_sq(self, u_num, u_ex):
        """|| grad_H(u_num) - grad_H(u_ex) ||_p^2, src/mms_trial_utils.py:101-106."""
        ex = self.dminus_x(u_num) - self.dminus_x(u_ex)
        ey = self.dminus_y(u_num) - self.dminus_y(u_ex)
        return self.normp_sq(ex, ey)


def shift(u, a, b):
    """v[i,j] = u[i+a, j+b], zero where out of range (src/prob1base.py:1553-1581)."""
    v = np.zeros_like(u)
    if a == 1:
        v[:-1, :] = u[1:, :]
    elif a == -1:
        v[1:, :] = u[:-1, :]
    elif b == 1:
        v[:, :-1] = u[:, 1:]
    elif b == -1:
        v[:, 1:] = u[:, :-1]
    else:
        v = u.copy()
    return v


# --------------------------------------------------------------------------------------
# MMS cases (closed forms)                       reference: src/prob1_mms_cases.py
# --------------------------------------------------------------------------------------
class MMSCase:
    """eval(t, X, Y) -> {var: {"u","ut","ux","uy","lap"}} for the five unknowns."""
    name = "base"

    def eval(self, t, X, Y):
        raise NotImplementedError

    def exact(self, t, X, Y):
        e = self.eval(t, X, Y)
        return {v: e[v]["u"] for v in VARS}


def _same_for_all(d):
    return {v: d for v in VARS}


class CasePol(MMSCase):  # src/prob1_mms_cases.py:250-277
    name = "pol"

    def __init__(self, model=None):
        pass

    def eval(self, t, X, Y):
        phi = 1.0 / (1.0 + t)
        dphi = -1.0 / (1.0 + t) ** 2
        fx, fy = X * (1 - X), Y * (1 - Y)
        dfx, dfy = 1 - 2 * X, 1 - 2 * Y
        d = dict(u=phi * fx * fy, ut=dphi * fx * fy, ux=phi * dfx * fy, uy=phi * fx * dfy,
                 lap=phi * (-2.0 * fy - 2.0 * fx))
        return _same_for_all(d)


class CasePeaks(MMSCase):  # src/prob1_mms_cases.py:151-247
    name = "peaks"

    def __init__(self, model=None, leading_spatial_const=1.0, evol_speed=1e1):
        self.C = float(leading_spatial_const)
        self.a = float(evol_speed)

    def eval(self, t, X, Y):
        pi = math.pi
        amp = self.C * math.exp(-self.a * t)
        sx, cx, sy, cy = np.sin(pi * X), np.cos(pi * X), np.sin(pi * Y), np.cos(pi * Y)
        R = X * X + Y * Y
        S = sx * sy
        R2, R3 = R * R, R * R * R
        u = amp * R3 * S
        ux = amp * (6 * X * R2 * S + R3 * pi * cx * sy)
        uy = amp * (6 * Y * R2 * S + R3 * pi * sx * cy)
        uxx = amp * (6 * R2 * S + 24 * X * X * R * S + 12 * X * R2 * pi * cx * sy - pi * pi * R3 * S)
        uyy = amp * (6 * R2 * S + 24 * Y * Y * R * S + 12 * Y * R2 * pi * sx * cy - pi * pi * R3 * S)
        d = dict(u=u, ut=-self.a * u, ux=ux, uy=uy, lap=uxx + uyy)
        return _same_for_all(d)


class CaseNonFullySmoothPol(MMSCase):  # src/prob1_mms_cases.py:406-511
    name = "nfs"

    def __init__(self, model=None, gamma=(2.1, 3.1), theta=1.0 / math.pi):
        g = [float(v) for v in np.atleast_1d(gamma)]
        if len(g) == 1:
            g = g * 5
        elif len(g) == 2:
            g = [g[0], g[1], g[1], g[1], g[0]]
        assert len(g) == 5
        self.gamma = g
        self.theta = float(theta)

    def _axis(self, X, gam):
        """f(x) = x(1-x)|x-theta|^gam and its first two derivatives (the Dirac term of
        the second derivative is dropped: it is multiplied by |x-theta|^(gam-1) = 0)."""
        d = X - self.theta
        ad = np.abs(d)
        sg = np.sign(d)
        p0 = ad ** gam
        p1 = gam * ad ** (gam - 1) * sg
        p2 = gam * (gam - 1) * ad ** (gam - 2)
        w, dw = X * (1 - X), 1 - 2 * X
        return w * p0, dw * p0 + w * p1, -2.0 * p0 + 2 * dw * p1 + w * p2

    def eval(self, t, X, Y):
        phi = 1.0 / (1.0 + t)
        dphi = -1.0 / (1.0 + t) ** 2
        out = {}
        for v, gam in zip(VARS, self.gamma):
            fx, dfx, ddfx = self._axis(X, gam)
            fy, dfy, ddfy = self._axis(Y, gam)
            out[v] = dict(u=phi * fx * fy, ut=dphi * fx * fy, ux=phi * dfx * fy,
                          uy=phi * fx * dfy, lap=phi * (ddfx * fy + fx * ddfy))
        return out


class CaseExpSin(MMSCase):  # src/prob1_mms_cases.py:280-337
    name = "expsin"

    def __init__(self, model):
        self.m = model

    def eval(self, t, X, Y):
        m = self.m
        pi = math.pi
        a = 2 * pi * pi * m.DT
        sx, cx, sy, cy = np.sin(pi * X), np.cos(pi * X), np.sin(pi * Y), np.cos(pi * Y)
        W, Wx, Wy = sx * sy, pi * cx * sy, pi * sx * cy
        lapW = -2 * pi * pi * W
        ea, e1, e2 = math.exp(-a * t), math.exp(-t), math.exp(-2 * t)
        em1a = math.expm1(-a * t) / a if a != 0 else -t  # (e^{-at}-1)/a
        em11 = math.expm1(-t)
        em12 = math.expm1(-2 * t)
        out = {}
        out["T"] = dict(u=ea * W, ut=-a * ea * W, ux=ea * Wx, uy=ea * Wy, lap=ea * lapW)
        out["cl"] = dict(u=-e1 * W, ut=e1 * W, ux=-e1 * Wx, uy=-e1 * Wy, lap=-e1 * lapW)
        out["cd"] = dict(u=e1 * W, ut=-e1 * W, ux=e1 * Wx, uy=e1 * Wy, lap=e1 * lapW)
        # cp = W exp(-K1 t + beta(t) W),  beta = -K1 (e^{-t}-1) + K2 (e^{-at}-1)/a
        beta = -m.K1 * em11 + m.K2 * em1a
        dbeta = m.K1 * e1 - m.K2 * ea
        E = np.exp(-m.K1 * t + beta * W)
        cp = W * E
        cpW = E * (1 + beta * W)
        out["cp"] = dict(u=cp, ut=cp * (-m.K1 + dbeta * W), ux=cpW * Wx, uy=cpW * Wy, lap=None)
        # cs = r_sp W exp(Q(t)-Q(0)),  Q' = -Kd (Sd - e^{-t}W)(1 - e^{-t}W)
        Qd = -m.Kd * (m.Sd * t + (m.Sd + 1) * em11 * W - 0.5 * em12 * W * W)
        dQ = -m.Kd * (m.Sd - e1 * W) * (1 - e1 * W)
        cs = m.r_sp * W * np.exp(Qd)
        out["cs"] = dict(u=cs, ut=cs * dQ, ux=None, uy=None, lap=None)
        return out


def make_case(name, model, **params):
    if name == "pol":
        return CasePol(model)
    if name == "expsin":
        return CaseExpSin(model)
    if name in ("peaks", "peaks_fast1e1"):
        params.setdefault("leading_spatial_const", 1.0)
        params.setdefault("evol_speed", 1e1)
        return CasePeaks(model, **params)
    if name in ("nfs", "nfs_h2_h3"):
        params.setdefault("gamma", (2.1, 3.1))
        return CaseNonFullySmoothPol(model, **params)
    raise ValueError(name)


# --------------------------------------------------------------------------------------
# MMS forcing terms                 reference: src/prob1base.py:2296-2425, 3468-3551, 493-598
# --------------------------------------------------------------------------------------
def forcing_pointwise(case, model, t, X, Y, react, eta, which=("fcp", "fT", "fcl", "fcd", "fcs")):
    e = case.eval(t, X, Y)
    cp, T, cl, cd, cs = (e[v]["u"] for v in VARS)
    m = model
    out = {}
    if "fcp" in which:
        out["fcp"] = e["cp"]["ut"] - (-cp * (m.K1 * (1 + cl) + m.K2 * T))
    if "fT" in which:
        out["fT"] = e["T"]["ut"] - (m.DT * e["T"]["lap"] - m.K3 * cp * T)
    if "fcl" in which:
        Dl, dDl = m.Dl(cp), m.dDl(cp)
        out["fcl"] = e["cl"]["ut"] - (
            dDl * (e["cp"]["ux"] * e["cl"]["ux"] + e["cp"]["uy"] * e["cl"]["uy"])
            + Dl * e["cl"]["lap"]
            - m.V1(T) * e["cl"]["ux"]
            - (cl + 1) * (m.gamma_T * e["T"]["ux"])
            - m.K4 * cp * (cl + 1))
    g = None
    if "fcd" in which or "fcs" in which:
        g = reaction_gain(cs, react, eta)
    if "fcd" in which:
        Dd, dc, dT = m.Dd(cp, T), m.dDd_dc(cp, T), m.dDd_dT(cp, T)
        out["fcd"] = e["cd"]["ut"] - (
            (dc * e["cp"]["ux"] + dT * e["T"]["ux"]) * e["cd"]["ux"]
            + (dc * e["cp"]["uy"] + dT * e["T"]["uy"]) * e["cd"]["uy"]
            + Dd * e["cd"]["lap"]
            + m.Kd * (m.Sd - cd) * (cl + 1) * g)
    if "fcs" in which:
        out["fcs"] = e["cs"]["ut"] - (-m.Kd * g * (1 + cl) * (m.Sd - cd))
    return out


_GL_NODES = np.array([-math.sqrt(3.0 / 5.0), 0.0, math.sqrt(3.0 / 5.0)])
_GL_WEIGHTS = np.array([5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0])


def cell_average(fn, mesh):
    """Average of fn over [x_{i-1/2}, x_{i+1/2}] x [y_{j-1/2}, y_{j+1/2}] for interior nodes by
    3x3 Gauss-Legendre; zero on the boundary (src/prob1base.py:493-598)."""
    x, y = mesh.x, mesh.y
    x_lo = 0.5 * (x[:-2] + x[1:-1])  # x_{i-1/2}, i = 1..N-1
    y_lo = 0.5 * (y[:-2] + y[1:-1])
    wx = mesh.hm[1:-1]
    wy = mesh.km[1:-1]
    acc = np.zeros((mesh.N - 1, mesh.M - 1))
    for a in range(3):
        p = x_lo + (_GL_NODES[a] + 1.0) * 0.5 * wx
        for b in range(3):
            q = y_lo + (_GL_NODES[b] + 1.0) * 0.5 * wy
            P, Q = np.meshgrid(p, q, indexing="ij")
            acc += _GL_WEIGHTS[a] * _GL_WEIGHTS[b] * fn(P, Q)
    out = mesh.zeros()
    out[1:-1, 1:-1] = 0.25 * acc
    return out


def forcing_arrays(case, model, mesh, t, react, eta):
    """The five forcing arrays the field uses at time t: cell-averaged fcp, nodal others."""
    f = forcing_pointwise(case, model, t, mesh.X, mesh.Y, react, eta,
                          which=("fT", "fcl", "fcd", "fcs"))
    f["fcp"] = cell_average(
        lambda P, Q: forcing_pointwise(case, model, t, P, Q, react, eta, which=("fcp",))["fcp"],
        mesh)
    return f


def zero_forcing(mesh):
    return {k: mesh.zeros() for k in ("fcp", "fT", "fcl", "fcd", "fcs")}


# --------------------------------------------------------------------------------------
# semi-discrete field F                           reference: src/prob1base.py:2489-2672
# --------------------------------------------------------------------------------------
def reaction_term(mesh, model, S, react, eta):
    g = reaction_gain(S["cs"], react, eta)
    return mesh.mask * ((1 + S["cl"]) * (model.Sd - S["cd"]) * (model.Kd * g))


def field_Fcp(mesh, model, S, f):
    out = f["fcp"].copy()
    r = -model.K1 * (S["cl"] + 1) * S["cp"] - model.K2 * S["T"] * S["cp"]
    out[1:-1, 1:-1] += r[1:-1, 1:-1]
    return out


def field_FT(mesh, model, S, f):
    out = f["fT"].copy()
    T = S["T"]
    r = (mesh.dstar_x(model.DT * mesh.dminus_x(T)) + mesh.dstar_y(model.DT * mesh.dminus_y(T))
         - model.K3 * S["cp"] * T)
    out[1:-1, 1:-1] += r[1:-1, 1:-1]
    return out


def field_Fcl(mesh, model, S, f):
    out = f["fcl"].copy()
    cp, T, cl = S["cp"], S["T"], S["cl"]
    flux_x = model.Dl(mesh.mean_x(cp)) * mesh.dminus_x(cl) - mesh.mean_x(model.V1(T) * (cl + 1))
    flux_y = model.Dl(mesh.mean_y(cp)) * mesh.dminus_y(cl)  # V2 == 0
    r = mesh.dstar_x(flux_x) + mesh.dstar_y(flux_y) - model.K4 * cp * (cl + 1)
    out[1:-1, 1:-1] += r[1:-1, 1:-1]
    return out


def field_Fcd(mesh, model, S, f, react, eta):
    out = f["fcd"].copy()
    cp, T, cd = S["cp"], S["T"], S["cd"]
    flux_x = model.Dd(mesh.mean_x(cp), mesh.mean_x(T)) * mesh.dminus_x(cd)
    flux_y = model.Dd(mesh.mean_y(cp), mesh.mean_y(T)) * mesh.dminus_y(cd)
    r = mesh.dstar_x(flux_x) + mesh.dstar_y(flux_y) + reaction_term(mesh, model, S, react, eta)
    out[1:-1, 1:-1] += r[1:-1, 1:-1]
    return out


def field_Fcs(mesh, model, S, f, react, eta):
    return (f["fcs"] - reaction_term(mesh, model, S, react, eta)) * mesh.mask


def field_all(mesh, model, S, f, react, eta):
    return {"cp": field_Fcp(mesh, model, S, f), "T": field_FT(mesh, model, S, f),
            "cl": field_Fcl(mesh, model, S, f), "cd": field_Fcd(mesh, model, S, f, react, eta),
            "cs": field_Fcs(mesh, model, S, f, react, eta)}


# --------------------------------------------------------------------------------------
# Jacobian stencils                               reference: src/prob1base.py:2511-2597, 2674-2839
# Each returns {"C","E","W","N","S"}: dF_ij/du_ij, /du_{i+1,j}, /du_{i-1,j}, /du_{i,j+1}, /du_{i,j-1};
# full-size arrays that vanish on the boundary.
# --------------------------------------------------------------------------------------
def _metric_weights(mesh):
    """Cx-,Cx+,Cy-,Cy+ = 1/(hm_i h_i), 1/(hm_i h_{i+1}), 1/(km_j k_j), 1/(km_j k_{j+1}) on the interior."""
    z = mesh.zeros
    cxm, cxp, cym, cyp = z(), z(), z(), z()
    cxm[1:-1, :] = 1.0 / (mesh.hmcol[1:-1, :] * mesh.hcol[1:-1, :])
    cxp[1:-1, :] = 1.0 / (mesh.hmcol[1:-1, :] * mesh.hcol[2:, :])
    cym[:, 1:-1] = 1.0 / (mesh.kmrow[:, 1:-1] * mesh.krow[:, 1:-1])
    cyp[:, 1:-1] = 1.0 / (mesh.kmrow[:, 1:-1] * mesh.krow[:, 2:])
    return cxm, cxp, cym, cyp


def _diffusion_stencil(mesh, Dx, Dy):
    """Stencil of u -> D*x(Dx D-x u) + D*y(Dy D-y u) with face coefficients Dx (face i-1|i
    stored at i) and Dy."""
    cxm, cxp, cym, cyp = _metric_weights(mesh)
    m = mesh.mask
    Dxp, Dyp = shift(Dx, 1, 0), shift(Dy, 0, 1)
    W = m * Dx * cxm
    E = m * Dxp * cxp
    S = m * Dy * cym
    Nn = m * Dyp * cyp
    return {"C": -(W + E + S + Nn), "E": E, "W": W, "N": Nn, "S": S}


def jac_T_FT(mesh, model, S):  # :2674-2684
    one = np.ones(mesh.shape)
    st = _diffusion_stencil(mesh, model.DT * one, model.DT * one)
    st["C"] = st["C"] - mesh.mask * model.K3 * S["cp"]
    return st


def _edge_drop(mesh, a, b):
    """1 on the interior, except 0 where node (i+a, j+b) is a boundary node (:1822-1851)."""
    d = mesh.mask.copy()
    if a == 1:
        d[mesh.N - 1, :] = 0
    elif a == -1:
        d[1, :] = 0
    elif b == 1:
        d[:, mesh.M - 1] = 0
    elif b == -1:
        d[:, 1] = 0
    return d


def jac_cl_Fcl(mesh, model, S):  # :2716-2750
    cp, T = S["cp"], S["T"]
    st = _diffusion_stencil(mesh, model.Dl(mesh.mean_x(cp)), model.Dl(mesh.mean_y(cp)))
    V1 = model.V1(T)
    half_over_hm = np.zeros(mesh.shape)
    half_over_hm[1:-1, :] = 0.5 / mesh.hmcol[1:-1, :]
    st["W"] = st["W"] + _edge_drop(mesh, -1, 0) * half_over_hm * shift(V1, -1, 0)
    st["E"] = st["E"] - _edge_drop(mesh, 1, 0) * half_over_hm * shift(V1, 1, 0)
    st["C"] = st["C"] - mesh.mask * model.K4 * cp
    return st


def jac_T_Fcl(mesh, model, S):  # :2686-2714
    cl = S["cl"]
    half_over_hm = np.zeros(mesh.shape)
    half_over_hm[1:-1, :] = 0.5 / mesh.hmcol[1:-1, :]
    z = mesh.zeros()
    W = _edge_drop(mesh, -1, 0) * (1 + shift(cl, -1, 0)) * model.gamma_T * half_over_hm
    E = -_edge_drop(mesh, 1, 0) * (1 + shift(cl, 1, 0)) * model.gamma_T * half_over_hm
    return {"C": z, "E": E, "W": W, "N": z.copy(), "S": z.copy()}


def jac_T_Fcd(mesh, model, S):  # :2752-2800   (reaction has no T dependence: a_T = 0)
    cp, T, cd = S["cp"], S["T"], S["cd"]
    gx = mesh.dminus_x(cd) * model.dDd_dT(mesh.mean_x(cp), mesh.mean_x(T))
    gy = mesh.dminus_y(cd) * model.dDd_dT(mesh.mean_y(cp), mesh.mean_y(T))
    m = mesh.mask
    hx = np.zeros(mesh.shape)
    hy = np.zeros(mesh.shape)
    hx[1:-1, :] = 0.5 / mesh.hmcol[1:-1, :]
    hy[:, 1:-1] = 0.5 / mesh.kmrow[:, 1:-1]
    W = -m * hx * gx
    E = m * hx * shift(gx, 1, 0)
    Sc = -m * hy * gy
    Nn = m * hy * shift(gy, 0, 1)
    return {"C": W + E + Sc + Nn, "E": E, "W": W, "N": Nn, "S": Sc}


def jac_cl_Fcd(mesh, model, S, react, eta):  # :2802-2809, :2541-2568
    g = reaction_gain(S["cs"], react, eta)
    z = mesh.zeros
    return {"C": mesh.mask * (model.Sd - S["cd"]) * (model.Kd * g), "E": z(), "W": z(), "N": z(), "S": z()}


def jac_cd_Fcd(mesh, model, S, react, eta):  # :2811-2839, :2570-2597
    cp, T = S["cp"], S["T"]
    st = _diffusion_stencil(mesh, model.Dd(mesh.mean_x(cp), mesh.mean_x(T)),
                            model.Dd(mesh.mean_y(cp), mesh.mean_y(T)))
    g = reaction_gain(S["cs"], react, eta)
    st["C"] = st["C"] - mesh.mask * (1 + S["cl"]) * (model.Kd * g)
    return st


def stencil_apply(mesh, st, w, interior_columns_only):
    """(J w) on interior rows.  interior_columns_only=True drops the boundary entries of w
    (delT_Fcl_at_w, :2234-2255); False keeps them (delT_Fcd_at_w/delcl_Fcd_at_w, :2257-2293)."""
    ww = w * mesh.mask if interior_columns_only else w
    out = (st["C"] * ww + st["E"] * shift(ww, 1, 0) + st["W"] * shift(ww, -1, 0)
           + st["N"] * shift(ww, 0, 1) + st["S"] * shift(ww, 0, -1))
    return out * mesh.mask


# --------------------------------------------------------------------------------------
# 5-diagonal solve                                reference: src/prob1base.py:601-710, 2088-2130
# --------------------------------------------------------------------------------------
def five_diag_matrix(mesh, C, E, W, Nn, S):
    """Sparse matrix over the interior unknowns (C-order), boundary columns dropped."""
    n1, m1 = mesh.N - 1, mesh.M - 1
    it = (slice(1, -1), slice(1, -1))
    c, e, w, n, s = (np.array(a[it]) for a in (C, E, W, Nn, S))
    e[-1, :] = 0.0
    w[0, :] = 0.0
    n[:, -1] = 0.0
    s[:, 0] = 0.0
    size = n1 * m1
    c, e, w, n, s = (a.reshape(size) for a in (c, e, w, n, s))
    diags = [c]
    offs = [0]
    if m1 < size:
        diags += [e[: size - m1], w[m1:]]
        offs += [m1, -m1]
    if size > 1:
        diags += [n[:-1], s[1:]]
        offs += [1, -1]
    return sp.diags(diags, offs, shape=(size, size), format="csc")


def solve_newton_system(mesh, st, dt, rhs_full, swap_ws=False):
    """Solve (2 I - dt J) delta = rhs on the interior; returns delta as a full array (zero
    boundary).  swap_ws reproduces the reference's cd assembly, where the (0,-1) coefficients
    are passed as the (i-1,j) diagonal and the (-1,0) ones as the (i,j-1) diagonal
    (src/prob1base.py:3097,3099)."""
    C = 2.0 - dt * st["C"]
    E, W, Nn, S = (-dt * st[k] for k in ("E", "W", "N", "S"))
    if swap_ws:
        W, S = S, W
    A = five_diag_matrix(mesh, C, E, W, Nn, S)
    sol = spla.spsolve(A, rhs_full[1:-1, 1:-1].reshape(-1))
    out = mesh.zeros()
    out[1:-1, 1:-1] = np.asarray(sol).reshape(mesh.N - 1, mesh.M - 1)
    return out


# --------------------------------------------------------------------------------------
# the problem bundle + integrators                reference: src/prob1base.py:2885-3219, 3596-3702
# --------------------------------------------------------------------------------------
@dataclass
class Problem:
    mesh: Mesh
    model: Model
    case: MMSCase | None
    react: str = "reg"       # "reg" (RegHCsTriple), "cs3" (CsTriple) or "hcs" (HCsTriple)
    eta: float = 50.0

    def forcing(self, t):
        if self.case is None:
            return zero_forcing(self.mesh)
        return forcing_arrays(self.case, self.model, self.mesh, t, self.react, self.eta)

    def exact_state(self, t):
        return self.case.exact(t, self.mesh.X, self.mesh.Y)


def step_forward_euler(pb, S, t0, dt, forcing0=None):  # :2885-2903
    f0 = pb.forcing(t0) if forcing0 is None else forcing0
    F = field_all(pb.mesh, pb.model, S, f0, pb.react, pb.eta)
    return {v: S[v] + dt * F[v] for v in VARS}


def cs_newton(y, a, eta, x0, iters, rtol):  # :3647-3663
    x = x0
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        for _ in range(iters):
            ex = np.exp(-eta * x)
            f = 2 * x + (2 * x - y) * ex - y + a
            J = 2 + 2 * ex - eta * (2 * x - y) * ex
            dx = -f / J
            x = x + dx
            if np.all(np.max(np.abs(dx)) < rtol * np.abs(x)):
                break
    return x


def step_pc(pb, S0, t0, dt, *, num_pc_steps=1, num_newton_steps=1, num_newton_iterations=5,
            consec_xs_rtol=1e-6, forcing0=None, forcing1=None, want_residual=True):
    """One predictor-corrector step (Heun predictor for cp, cs; one Newton step of the
    trapezoidal rule for T -> cl -> cd; trapezoidal corrector for cp, cs),
    src/prob1base.py:3117-3149.  Returns (S1, residuals)."""
    assert dt > 0
    mesh, m, react, eta = pb.mesh, pb.model, pb.react, pb.eta
    t1 = t0 + dt
    f0 = pb.forcing(t0) if forcing0 is None else forcing0
    f1 = pb.forcing(t1) if forcing1 is None else forcing1
    mask = mesh.mask

    F0 = field_all(mesh, m, S0, f0, react, eta)
    YT = dt * F0["T"] + 2 * S0["T"]
    Ycl = dt * F0["cl"] + 2 * S0["cl"]
    Ycd = dt * F0["cd"] + 2 * S0["cd"]

    # Heun predictors (:2953-2965, :3631-3645, :3174-3188)
    cp_star = S0["cp"] + dt * F0["cp"]
    Fcp_star = field_Fcp(mesh, m, dict(S0, cp=cp_star), f1)
    cp1 = S0["cp"] + 0.5 * dt * (F0["cp"] + Fcp_star)
    cs_star = S0["cs"] + dt * F0["cs"]
    Fcs_star = field_Fcs(mesh, m, dict(S0, cs=cs_star), f1, react, eta)
    cs1 = S0["cs"] + (0.5 * dt) * (F0["cs"] + Fcs_star)
    if react in ("reg", "hcs"):                                       # :3645, :3374
        cs1 = cs1 * mask

    T1, cl1, cd1 = S0["T"], S0["cl"], S0["cd"]
    res = {}
    for _pc in range(num_pc_steps):
        for _nw in range(num_newton_steps):
            Sp = {"cp": cp1, "T": T1, "cl": cl1, "cd": cd1, "cs": cs1}
            # --- T (:2998-3045)
            rhs = YT - (2 * Sp["T"] - dt * field_FT(mesh, m, Sp, f1))
            dT = solve_newton_system(mesh, jac_T_FT(mesh, m, Sp), dt, rhs)
            Tn = mesh.zeros()
            Tn[1:-1, 1:-1] = Sp["T"][1:-1, 1:-1] + dT[1:-1, 1:-1]
            if want_residual:
                res["T"] = 2 * Tn - dt * field_FT(mesh, m, dict(Sp, T=Tn), f1) - YT
            # --- cl (:3047-3080)
            rhs = (Ycl - 2 * Sp["cl"] + dt * field_Fcl(mesh, m, Sp, f1)
                   + dt * stencil_apply(mesh, jac_T_Fcl(mesh, m, Sp), Tn - Sp["T"], True))
            dcl = solve_newton_system(mesh, jac_cl_Fcl(mesh, m, Sp), dt, rhs)
            cln = Sp["cl"] + dcl
            if want_residual:
                res["cl"] = 2 * cln - dt * field_Fcl(mesh, m, dict(Sp, T=Tn, cl=cln), f1) - Ycl
            # --- cd (:3082-3115)
            rhs = (Ycd - 2 * Sp["cd"] + dt * field_Fcd(mesh, m, Sp, f1, react, eta)
                   + dt * stencil_apply(mesh, jac_T_Fcd(mesh, m, Sp), Tn - Sp["T"], False)
                   + dt * stencil_apply(mesh, jac_cl_Fcd(mesh, m, Sp, react, eta), cln - Sp["cl"], False))
            dcd = solve_newton_system(mesh, jac_cd_Fcd(mesh, m, Sp, react, eta), dt, rhs, swap_ws=True)
            cdn = Sp["cd"] + dcd
            if want_residual:
                res["cd"] = (2 * cdn - dt * field_Fcd(mesh, m, dict(Sp, T=Tn, cl=cln, cd=cdn), f1, react, eta)
                             - Ycd)
            T1, cl1, cd1 = Tn, cln, cdn
        # --- correctors
        it = (slice(1, -1), slice(1, -1))
        a0 = -m.K2 * S0["T"][it] - m.K1 * (S0["cl"][it] + 1)          # :2967-2996
        a1 = -m.K2 * T1[it] - m.K1 * (cl1[it] + 1)
        num = (1 + (dt / 2) * a0) * S0["cp"][it] + (dt / 2) * (f0["fcp"][it] + f1["fcp"][it])
        cp1 = mesh.zeros()
        cp1[it] = num / (1 - (dt / 2) * a1)
        if react == "reg":                                            # :3665-3702
            H0 = heaviside_reg(S0["cs"], eta)
            y = (2 * S0["cs"] - dt * m.Kd * (m.Sd - S0["cd"]) * (S0["cl"] + 1) * H0
                 + dt * (f0["fcs"] + f1["fcs"]))
            a = dt * m.Kd * (m.Sd - cd1) * (cl1 + 1)
            cs1 = cs_newton(y, a, eta, S0["cs"], num_newton_iterations, consec_xs_rtol) * mask
        elif react == "hcs":                                          # :3376-3430
            tol = np.finfo(float).eps * 100
            del_Y1 = 2 - dt * (m.Sd - cd1) * (1 + cl1) * m.Kd
            if np.any(del_Y1 < tol):
                raise ValueError("Denominator 2 - dt Kd (Sd - Cd1) (1 + Cl1) below positiveness treshold.")
            Y0 = 2 * S0["cs"] + dt * F0["cs"] + dt * f1["fcs"]
            cs1 = mesh.zeros()
            pos, neg = Y0 > tol, Y0 < -tol
            cs1[pos] = Y0[pos] / del_Y1[pos]
            cs1[neg] = Y0[neg] / 2.0
            cs1 = cs1 * mask
        else:                                                         # :3190-3219
            b0 = -m.Kd * (m.Sd - S0["cd"][it]) * (1 + S0["cl"][it])
            b1 = -m.Kd * (m.Sd - cd1[it]) * (1 + cl1[it])
            num = (1 + (dt / 2) * b0) * S0["cs"][it] + (dt / 2) * (f0["fcs"][it] + f1["fcs"][it])
            cs1 = mesh.zeros()
            cs1[it] = num / (1 - (dt / 2) * b1)
    return {"cp": cp1, "T": T1, "cl": cl1, "cd": cd1, "cs": cs1}, res


# --------------------------------------------------------------------------------------
# MMS error harness                               reference: src/mms_trial_utils.py:15-147
# --------------------------------------------------------------------------------------
INTEGRAL_VARS = ("T", "cl", "cd")


def error_record(pb, S, t):
    ex = pb.exact_state(t)
    hsq = {v: pb.mesh.normH_sq(S[v] - ex[v]) for v in VARS}
    gsq = {v: (pb.mesh.grad_err_sq(S[v], ex[v]) if v in INTEGRAL_VARS else 0.0) for v in VARS}
    return hsq, gsq


def combined_error(series, dt, integral_vars, all_vars=None):
    """max_k sqrt( sum_v ||e_v(t_k)||_H^2 + trapz_0^{t_k} sum_{v in integral} ||grad e_v||_p^2 )."""
    integrand = [sum(g[v] for v in integral_vars) for (_h, g) in series]
    best, running = 0.0, 0.0
    for k, (h, _g) in enumerate(series):
        hs = sum(h.values()) if all_vars is None else sum(h[v] for v in all_vars)
        if k > 0:
            running += 0.5 * dt * (integrand[k - 1] + integrand[k])
        best = max(best, hs + running)
    return math.sqrt(best)


def run_for_errors(pb, Tf, dt, t0=0.0, integrator="pc", **integ_kw):
    """Restatement of MMSTrial.run_for_errors (src/mms_trial_utils.py:254-280)."""
    nsteps = math.ceil((Tf - t0) / dt)
    dt = (Tf - t0) / nsteps
    S = pb.exact_state(t0)
    t = t0
    series = [error_record(pb, S, t)]
    f_prev = pb.forcing(t)
    for _ in range(nsteps):
        f_next = pb.forcing(t + dt)
        if integrator == "pc":
            S, _ = step_pc(pb, S, t, dt, forcing0=f_prev, forcing1=f_next, want_residual=False,
                           **integ_kw)
        else:
            S = step_forward_euler(pb, S, t, dt, forcing0=f_prev)
        t += dt
        f_prev = f_next
        series.append(error_record(pb, S, t))
    overall = combined_error(series, dt, INTEGRAL_VARS)
    per_var = {v: combined_error(series, dt, [v] if v in INTEGRAL_VARS else [], [v]) for v in VARS}
    return {"dt_used": dt, "overall": overall, "per_variable": per_var, "final_state": S,
            "series": series}
