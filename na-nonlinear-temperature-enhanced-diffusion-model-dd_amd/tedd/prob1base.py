"""Host-side mirror of the reference's solver API (src/prob1base.py) over the HIP path.

Same class names, constructor arguments, attribute names and error behaviour as the reference,
so its notebooks and tests can run on this module unchanged:

    import tedd.prob1base as p1

What runs where
  * The time-stepping hot path -- `integrator.step`, the field evaluations `F*`, the Jacobian
    stencils, the MMS forcing terms and exact solutions -- runs on the GPU through the C ABI
    (`tedd._ffi`).  There is no CPU fallback: without the HIP library these calls raise.
  * `Grid`, `StateVars`, the model classes and the 1-D difference/mean operators keep their
    NumPy form: they are containers and array utilities that callers apply to arbitrary arrays.

Reference file:line citations are given per class.
"""
from __future__ import annotations

import math
from abc import ABC, abstractmethod
from collections import namedtuple
from collections.abc import Mapping
from typing import Callable, Dict, List, Literal, NamedTuple

import numpy as np

from . import _ffi
from . import _symprog

Diag5 = namedtuple("Diag5", ["sub", "rspec", "cspec"])


# ------------------------------------------------------------------------------------------
# model constants and constitutive laws                     reference src/prob1base.py:28-217
# ------------------------------------------------------------------------------------------
class ModelConsts(NamedTuple):
    R0: float
    Ea: float
    K1: float
    K2: float
    K3: float
    K4: float
    DT: float
    Dl_max: float
    phi_l: float
    gamma_T: float
    Kd: float
    Sd: float
    Dd_max: float
    phi_d: float
    phi_T: float
    r_sp: float
    T_ref: float = 300


R0 = 8.3144621
Ea = 1.60217662e-19
default_model_consts = ModelConsts(
    R0=R0, Ea=Ea, K1=1e-2, K2=1e-2, K3=1e-2, K4=1e-2, DT=1e-3, Dl_max=8.01e-4, phi_l=1e-5,
    gamma_T=1e-9, Kd=1e-8, Sd=10, Dd_max=2.46e-6, phi_d=1e-5, phi_T=Ea / R0, r_sp=5e-2, T_ref=300)


class DefaultModel01:
    """Dl(cp) = Dl_max e^{-phi_l cp}; V1(T) = gamma_T T, V2 = 0; Dd(cp,T) = Dd_max e^{-phi_d cp}
    e^{-phi_T/T} with Dd := 0 where T == 0 (reference src/prob1base.py:71-202).  Constants are plain
    attributes and may be mutated in place; the integrators re-read them at every step."""

    model_kind = 1

    def __init__(self, mc: ModelConsts):
        for k, v in mc._asdict().items():
            setattr(self, k, v)

    def consts_vector(self):
        return [float(getattr(self, k)) for k in ModelConsts._fields]

    def with_changes(self, **kwargs):
        out = type(self)(ModelConsts(**{k: getattr(self, k) for k in ModelConsts._fields}))
        for k, v in kwargs.items():
            setattr(out, k, v)
        return out

    def copy(self):
        return self.with_changes()

    def Dl(self, cp, *, d=0):
        return ((-self.phi_l) ** d) * (self.Dl_max * np.exp(-self.phi_l * np.asarray(cp, dtype=float)))

    def V1(self, T, *, d=0):
        T = np.asarray(T, dtype=float)
        if d == 0:
            return self.gamma_T * T
        if d == 1:
            return self.gamma_T * np.ones_like(T)
        return np.zeros_like(T)

    def V2(self, T, *, d=0):
        return np.zeros_like(np.asarray(T, dtype=float))

    def _Dd_at(self, cp, Te, d):
        cp = np.asarray(cp, dtype=float)
        Te = np.asarray(Te, dtype=float)
        assert cp.shape == Te.shape
        nz = Te != 0
        out = np.zeros_like(Te, dtype=np.float64)
        out[nz] = self.Dd_max * np.exp(-self.phi_d * cp[nz]) * np.exp(-self.phi_T / Te[nz])
        if d == (0, 0):
            return out
        if d == (1, 0):
            return -self.phi_d * out
        if d == (0, 1):
            out[nz] *= self.phi_T / (Te[nz] ** 2)
            return out
        raise ValueError("derivative order not supported")

    def Dd(self, cp, T, *, d=(0, 0)):
        return self._Dd_at(cp, T, tuple(d))


class DefaultModel02(DefaultModel01):
    """DefaultModel01 with Dd evaluated at T + T_ref (reference src/prob1base.py:205-217)."""

    model_kind = 2

    def Dd(self, cp, T, *, d=(0, 0)):
        return self._Dd_at(cp, np.asarray(T, dtype=float) + self.T_ref, tuple(d))


# ------------------------------------------------------------------------------------------
# grid and 1-D operators                              reference src/prob1base.py:220-490, 1499-1581
# ------------------------------------------------------------------------------------------
def Mx_reg(u):
    out = np.zeros_like(u)
    out[1:, :] = 0.5 * (u[1:, :] + u[:-1, :])
    return out


def My_reg(u):
    out = np.zeros_like(u)
    out[:, 1:] = 0.5 * (u[:, 1:] + u[:, :-1])
    return out


def Dx_reg(u, hh):
    out = np.zeros_like(u)
    out[1:, :] = (u[1:, :] - u[:-1, :]) / hh[1:, :]
    return out


def Dy_reg(u, kk):
    out = np.zeros_like(u)
    out[:, 1:] = (u[:, 1:] - u[:, :-1]) / kk[:, 1:]
    return out


def Dx_star(u, hh_phalf):
    out = np.zeros_like(u)
    out[1:-1, :] = (u[2:, :] - u[1:-1, :]) / hh_phalf[1:-1, :]
    return out


def Dy_star(u, kk_phalf):
    out = np.zeros_like(u)
    out[:, 1:-1] = (u[:, 2:] - u[:, 1:-1]) / kk_phalf[:, 1:-1]
    return out


def grid0_fn_ab(u, a, b):
    assert a in (-1, 0, 1) and b in (-1, 0, 1) and (a == 0 or b == 0)
    v = np.zeros_like(u)
    if a == 0 and b == 0:
        v = np.copy(u)
    elif a == 1:
        v[:-1, :] = u[1:, :]
    elif a == -1:
        v[1:, :] = u[:-1, :]
    elif b == 1:
        v[:, :-1] = u[:, 1:]
    else:
        v[:, 1:] = u[:, :-1]
    return v


class Grid:
    """Tensor-product grid x[0..N] x y[0..M] with the reference's metric arrays
    (src/prob1base.py:220-362): h[0] = k[0] = inf, h_phalf[0] = h_phalf[N] = inf.  The 2-D arrays
    (xx, yy, hh, kk, ...) are built on first access, so a very large grid costs nothing until a
    caller asks for them."""

    def __init__(self, x: np.ndarray, y: np.ndarray):
        x = np.asarray(x, dtype=float)
        y = np.asarray(y, dtype=float)
        assert x.ndim == y.ndim and x.ndim in (1, 2), "Grid: x,y's shape must be 1D or 2D."
        if x.ndim == 2:
            assert x.shape == y.shape, "Grid: for meshgrid'ed x,y, same shape is required."
            x, y = x[:, 0], y[0, :]
        self.x, self.y = x, y
        self.N, self.M = len(x) - 1, len(y) - 1
        inf = np.array([np.inf])
        self.h = np.concatenate([inf, x[1:] - x[:-1]])
        self.k = np.concatenate([inf, y[1:] - y[:-1]])
        self.h_phalf = np.concatenate([(self.h[:-1] + self.h[1:]) * 0.5, inf])
        self.k_phalf = np.concatenate([(self.k[:-1] + self.k[1:]) * 0.5, inf])
        self._lazy = {}

    def _get(self, name, make):
        if name not in self._lazy:
            self._lazy[name] = make()
        return self._lazy[name]

    xx = property(lambda s: s._get("xx", lambda: np.meshgrid(s.x, s.y, indexing="ij")[0]))
    yy = property(lambda s: s._get("yy", lambda: np.meshgrid(s.x, s.y, indexing="ij")[1]))
    hh = property(lambda s: s._get("hh", lambda: np.meshgrid(s.k, s.h)[1]))
    kk = property(lambda s: s._get("kk", lambda: np.meshgrid(s.k, s.h)[0]))
    hh_phalf = property(lambda s: s._get("hh_phalf", lambda: np.meshgrid(s.k_phalf, s.h_phalf)[1]))
    kk_phalf = property(lambda s: s._get("kk_phalf", lambda: np.meshgrid(s.k_phalf, s.h_phalf)[0]))

    @property
    def xx_phalf(self):
        def make():
            out = np.zeros(self.full_shape)
            out[:-1, :] = 0.5 * (self.xx[:-1, :] + self.xx[1:, :])
            return out
        return self._get("xx_phalf", make)

    @property
    def yy_phalf(self):
        def make():
            out = np.zeros(self.full_shape)
            out[:, :-1] = 0.5 * (self.yy[:, :-1] + self.yy[:, 1:])
            return out
        return self._get("yy_phalf", make)

    @property
    def d5(self):
        def make():
            N, M = self.N, self.M
            jj, ii = np.meshgrid(np.arange(1, M), np.arange(1, N))
            sub = np.reshape(ii * (M + 1) + jj, (N - 1) * (M - 1))
            return Diag5(sub=sub, rspec=np.concatenate([sub] * 5),
                         cspec=np.concatenate([sub, sub - 1, sub + 1, sub - 1 - M, sub + 1 + M]))
        return self._get("d5", make)

    @property
    def full_shape(self):
        return (self.N + 1, self.M + 1)

    @property
    def interior_shape(self):
        return (self.N - 1, self.M - 1)

    def make_full0(self):
        return np.zeros(self.full_shape)

    def const_with_nullbd(self, v):
        out = np.zeros(self.full_shape)
        out[1:-1, 1:-1] = v
        return out

    @property
    def null_bd_mask(self):
        return self._get("mask", lambda: self.const_with_nullbd(1))

    # discrete inner products and norms (src/prob1base.py:387-433)
    def inner_product_H(self, u, v):
        return np.sum(u[1:-1, 1:-1] * np.conjugate(v[1:-1, 1:-1])
                      * self.h_phalf[1:-1, None] * self.k_phalf[None, 1:-1])

    def norm_H(self, u):
        return np.sqrt(self.inner_product_H(u, u))

    def inner_product_pk(self, u, v):
        return np.sum(u[1:, 1:-1] * np.conjugate(v[1:, 1:-1]) * self.h[1:, None] * self.k_phalf[None, 1:-1])

    def norm_pk(self, u):
        return np.sqrt(self.inner_product_pk(u, u))

    def inner_product_hp(self, u, v):
        return np.sum(u[1:-1, 1:] * np.conjugate(v[1:-1, 1:]) * self.h_phalf[1:-1, None] * self.k[None, 1:])

    def norm_hp(self, u):
        return np.sqrt(self.inner_product_hp(u, u))

    def inner_product_p(self, ux, uy, vx, vy):
        return self.inner_product_pk(ux, vx) + self.inner_product_hp(uy, vy)

    def norm_p(self, ux, uy):
        return np.sqrt(self.inner_product_p(ux, uy, ux, uy))

    def Dx_reg(self, u):
        return Dx_reg(u, self.hh)

    def Dy_reg(self, u):
        return Dy_reg(u, self.kk)

    def Dx_star(self, u):
        return Dx_star(u, self.hh_phalf)

    def Dy_star(self, u):
        return Dy_star(u, self.kk_phalf)

    def grad_H(self, u):
        return (self.Dx_reg(u), self.Dy_reg(u))


def make_uniform_grid(N: int, M: int):
    return Grid(np.linspace(0, 1, N + 1), np.linspace(0, 1, M + 1))


# ------------------------------------------------------------------------------------------
# StateVars                                               reference src/prob1base.py:1854-2085
# ------------------------------------------------------------------------------------------
_DERIVED = {
    "MxT": lambda s: Mx_reg(s.T), "MyT": lambda s: My_reg(s.T),
    "Mxcp": lambda s: Mx_reg(s.cp), "Mycp": lambda s: My_reg(s.cp),
    "DmxT": lambda s: Dx_reg(s.T, s.hh), "DmyT": lambda s: Dy_reg(s.T, s.kk),
    "Dmxcl": lambda s: Dx_reg(s.cl, s.hh), "Dmycl": lambda s: Dy_reg(s.cl, s.kk),
    "Dmxcd": lambda s: Dx_reg(s.cd, s.hh), "Dmycd": lambda s: Dy_reg(s.cd, s.kk),
    "Dl_Mxcp": lambda s: s.model.Dl(s.Mxcp), "Dl_Mycp": lambda s: s.model.Dl(s.Mycp),
    "dDl_Mxcp": lambda s: s.model.Dl(s.Mxcp, d=1), "dDl_Mycp": lambda s: s.model.Dl(s.Mycp, d=1),
    "V1T": lambda s: s.model.V1(s.T), "V2T": lambda s: s.model.V2(s.T),
    "dV1T": lambda s: s.model.V1(s.T, d=1), "dV2T": lambda s: s.model.V2(s.T, d=1),
    "Dd_MxcpT": lambda s: s.model.Dd(s.Mxcp, s.MxT), "Dd_MycpT": lambda s: s.model.Dd(s.Mycp, s.MyT),
    "delcp_Dd_MxcpT": lambda s: s.model.Dd(s.Mxcp, s.MxT, d=(1, 0)),
    "delcp_Dd_MycpT": lambda s: s.model.Dd(s.Mycp, s.MyT, d=(1, 0)),
    "delT_Dd_MxcpT": lambda s: s.model.Dd(s.Mxcp, s.MxT, d=(0, 1)),
    "delT_Dd_MycpT": lambda s: s.model.Dd(s.Mycp, s.MyT, d=(0, 1)),
}


class StateVars:
    """Immutable container of the five grid functions (cp, T, cl, cd, cs) plus `model`, `hh`, `kk`.
    The 24 derived arrays of the reference (face means, backward differences, Dl/V/Dd at the
    faces) are still offered as lazily cached read-only properties for callers that inspect
    them; the device path never materialises them (it recomputes face values in registers)."""

    _COMPUTED_PROPERTIES = _DERIVED

    def __init__(self, cp, T, cl, cd, cs, *, model, hh, kk):
        put = object.__setattr__

        def frozen(a):
            # The device copy of a state and the field values computed from it are cached by the identity
            # of the StateVars: an in-place edit of state.T[...] would go unseen, so it is made to fail.
            a = np.asarray(a)
            if not a.flags.writeable:
                return a
            v = a.view()
            v.setflags(write=False)
            return v
        put(self, "_cp_data", frozen(cp))
        put(self, "_T_data", frozen(T))
        put(self, "_cl_data", frozen(cl))
        put(self, "_cd_data", frozen(cd))
        put(self, "_cs_data", frozen(cs))
        put(self, "_model", model)
        put(self, "_hh", hh)
        put(self, "_kk", kk)
        put(self, "_initialized", True)

    cp = property(lambda s: s._cp_data)
    T = property(lambda s: s._T_data)
    cl = property(lambda s: s._cl_data)
    cd = property(lambda s: s._cd_data)
    cs = property(lambda s: s._cs_data)
    model = property(lambda s: s._model)
    hh = property(lambda s: s._hh)
    kk = property(lambda s: s._kk)

    def __getattr__(self, name):
        fn = _DERIVED.get(name)
        if fn is None:
            raise AttributeError(name)
        value = fn(self)
        object.__setattr__(self, "_cache_" + name, value)
        object.__setattr__(self, name, value)
        return value

    def __setattr__(self, name, value):
        if name.startswith("_cache_") or not self.__dict__.get("_initialized", False):
            object.__setattr__(self, name, value)
        else:
            raise AttributeError(f"Cannot set attribute '{name}'. '{type(self).__name__}' instance is immutable.")

    def __delattr__(self, name):
        if name.startswith("_cache_") or not self.__dict__.get("_initialized", False):
            object.__delattr__(self, name)
        else:
            raise AttributeError(f"Cannot delete attribute '{name}'. '{type(self).__name__}' instance is immutable.")

    def into_dict(self, recipient: dict, which: List[str] = None):
        names = list(which) if which is not None else list(_DERIVED) + ["cp", "T", "cl", "cd", "cs"]
        for n in names:
            recipient[n] = getattr(self, n)
        return recipient

    def with_changes(self, **kwargs):
        allowed = ["cp", "T", "cl", "cd", "cs"]
        current = {k: getattr(self, k) for k in allowed}
        for k, v in kwargs.items():
            if k not in allowed:
                raise ValueError(f"{k}: Invalid change. Can only change: {allowed}.")
            current[k] = v
        return StateVars(current["cp"], current["T"], current["cl"], current["cd"], current["cs"],
                         model=self.model, hh=self.hh, kk=self.kk)

    def copy(self):
        return self.with_changes()


# ------------------------------------------------------------------------------------------
# device sessions
# ------------------------------------------------------------------------------------------
class _Session:
    """One C-ABI context for a (grid, reaction kind, forcing source) combination, with the model
    constants re-sent whenever they changed since the last call."""

    def __init__(self, grid: Grid, model, react, eta, mms_case, mms_params, program=None):
        self.grid = grid
        self.react = react
        self.mms_case = mms_case
        self._sent = None
        self.ctx = _ffi.Context(grid.x, grid.y, model.consts_vector(), model_kind=model.model_kind,
                                reaction_kind=react, eta=eta, mms_case=mms_case, mms_params=mms_params)
        if mms_case == _ffi.MMS_PROGRAM:
            node, cell = program
            self.ctx.set_mms_program(0, *node)
            self.ctx.set_mms_program(1, *cell)
        self._sent = self._key(model, eta, mms_params)
        self.state_token = None   # identity of the StateVars the device state equals

    @staticmethod
    def _key(model, eta, mms_params):
        return (tuple(model.consts_vector()), model.model_kind, float(eta), tuple(float(p) for p in mms_params))

    def refresh(self, model, eta, mms_params):
        key = self._key(model, eta, mms_params)
        if key != self._sent:
            self.ctx.set_model(model.consts_vector(), model.model_kind, eta, mms_params)
            self._sent = key

    def load_state(self, state: StateVars):
        if self.state_token is not None and self.state_token is state:
            return
        self.ctx.set_state(state.cp, state.T, state.cl, state.cd, state.cs)
        self.state_token = state


def _same_nodes(grid: Grid, xx, yy):
    """True if (xx, yy) are the grid's own node coordinates."""
    if xx is None and yy is None:
        return True
    if xx is grid._lazy.get("xx") and yy is grid._lazy.get("yy"):
        return True
    xx, yy = np.asarray(xx), np.asarray(yy)
    return (xx.shape == grid.full_shape and yy.shape == grid.full_shape
            and np.array_equal(xx[:, 0], grid.x) and np.array_equal(yy[0, :], grid.y)
            and np.array_equal(xx[:, -1], grid.x) and np.array_equal(yy[-1, :], grid.y))


class _MeshSessions:
    """Device sessions of an object that is asked to evaluate at node coordinates (xx, yy): its own grid,
    or any other tensor-product mesh handed in as meshgrid arrays (the reference's cases and forcing terms
    are plain functions of the coordinates, e.g. a case built on one grid and evaluated on a family of
    refined grids).  A few foreign meshes are kept, the least recently used dropped."""

    KEEP = 4

    def __init__(self):
        self._foreign = {}    # bytes of the axes -> [Grid, session or None]

    def grid_for(self, own: Grid, xx, yy):
        if _same_nodes(own, xx, yy):
            return own
        xx, yy = np.asarray(xx, dtype=float), np.asarray(yy, dtype=float)
        if not (xx.ndim == 2 and xx.shape == yy.shape and xx.shape[0] >= 3 and xx.shape[1] >= 3
                and np.array_equal(xx, np.broadcast_to(xx[:, :1], xx.shape))
                and np.array_equal(yy, np.broadcast_to(yy[:1, :], yy.shape))
                and np.all(np.diff(xx[:, 0]) > 0) and np.all(np.diff(yy[0, :]) > 0)):
            raise NotImplementedError("the device library evaluates at the nodes of a tensor-product mesh "
                                      "(meshgrid arrays of increasing axes with at least 3 nodes each)")
        key = xx[:, 0].tobytes() + b"|" + yy[0, :].tobytes()
        entry = self._foreign.pop(key, None)
        if entry is None:
            entry = [Grid(xx[:, 0].copy(), yy[0, :].copy()), None]
            while len(self._foreign) >= self.KEEP:
                self._foreign.pop(next(iter(self._foreign)))
        self._foreign[key] = entry     # most recently used last
        return entry[0]

    def session(self, own: Grid, grid: Grid, own_session, make):
        """The session on `grid`; `own_session` is the caller's session on its own grid (or None)."""
        if grid is own:
            return own_session if own_session is not None else make(grid)
        for entry in self._foreign.values():
            if entry[0] is grid:
                if entry[1] is None:
                    entry[1] = make(grid)
                return entry[1]
        return make(grid)


# ------------------------------------------------------------------------------------------
# MMS cases and forcing terms          reference src/prob1base.py:714-889, 2296-2425, 3468-3551
# ------------------------------------------------------------------------------------------
class MMSCaseBase(ABC):
    def __init__(self, grid: Grid, model: DefaultModel01):
        self._model = model
        self._grid = grid

    @property
    def grid(self):
        return self._grid

    @property
    def model(self):
        return self._model


_WHAT = {"": 0, "dt_": 1, "dx_": 2, "dy_": 3, "lap_": 4}
_VIDX = {"cp": 0, "T": 1, "cl": 2, "cd": 3, "cs": 4}


class MMSCaseDevice(MMSCaseBase):
    """An MMS case whose exact solution is built into the device library (closed form).  The
    accessors cp/T/cl/cd/cs, dt_*, dx_*, dy_*, lap_* evaluate on the GPU at the grid nodes."""

    mms_case_id = None

    def __init__(self, grid: Grid, model: DefaultModel01):
        super().__init__(grid, model)
        self._session = None
        self._meshes = _MeshSessions()
        self._memo = {}

    def mms_params(self):
        return ()

    def _make_session(self, grid):
        return _Session(grid, self._model, _ffi.REACT_REGHCSTRIPLE, 0.0, self.mms_case_id, self.mms_params())

    def _device(self, grid=None):
        grid = self._grid if grid is None else grid
        s = self._meshes.session(self._grid, grid, self._session, self._make_session)
        if grid is self._grid:
            self._session = s
        s.refresh(self._model, 0.0, self.mms_params())
        return s

    def _eval(self, what, t, xx, yy):
        try:
            grid = self._meshes.grid_for(self._grid, xx, yy)
        except NotImplementedError:
            return self._eval_points(what, t, xx, yy)
        key = (what, float(t), tuple(self._model.consts_vector()), id(grid))
        if key not in self._memo:
            self._memo.clear()
            self._memo[key] = self._device(grid).ctx.eval_mms(t, what)
            self._memo["grid"] = grid      # keeps id(grid) unambiguous while the entry lives
        return self._memo[key]

    _MAX_POINT_MESH = 1 << 22

    def _eval_points(self, what, t, xx, yy):
        """Values at points that are not the nodes of a tensor-product mesh (a scalar pair, a line of points,
        ...), as the reference's cases allow: the device evaluates on the mesh of the distinct coordinates and
        the points are picked out of it."""
        px, py = np.broadcast_arrays(np.asarray(xx, dtype=float), np.asarray(yy, dtype=float))
        xs, ix = np.unique(px.ravel(), return_inverse=True)
        ys, iy = np.unique(py.ravel(), return_inverse=True)

        def padded(axis):       # a device mesh needs three increasing coordinates per axis
            extra = [axis[-1] + k for k in range(1, 4 - len(axis))] if len(axis) < 3 else []
            return np.concatenate([axis, np.asarray(extra, dtype=float)])
        xs, ys = padded(xs), padded(ys)
        if not (np.all(np.isfinite(xs)) and np.all(np.isfinite(ys))) or len(xs) * len(ys) > self._MAX_POINT_MESH:
            raise NotImplementedError("too many distinct coordinates for pointwise evaluation on the device; "
                                      "hand over the nodes of a tensor-product mesh")
        mx, my = np.meshgrid(xs, ys, indexing="ij")
        values = self._eval(what, t, mx, my)
        return [np.asarray(v)[ix.ravel(), iy.ravel()].reshape(px.shape) for v in values]

    def exact_fields(self, t):
        return [a.copy() for a in self._eval(0, t, None, None)]


def _add_mms_accessors(cls):
    for prefix, what in _WHAT.items():
        for var, idx in _VIDX.items():
            if (what == 4 and var in ("cp", "cs")) or (what in (2, 3) and var == "cs"):
                continue
            def method(self, t, xx=None, yy=None, _w=what, _i=idx):
                return self._eval(_w, t, xx, yy)[_i].copy()
            method.__name__ = prefix + var
            setattr(cls, prefix + var, method)
    return cls


_add_mms_accessors(MMSCaseDevice)


class ForcingTermsBase(ABC):
    @abstractmethod
    def fcp(self, t, xx, yy) -> np.ndarray:
        pass

    @abstractmethod
    def fT(self, t, xx, yy) -> np.ndarray:
        pass

    @abstractmethod
    def fcl(self, t, xx, yy) -> np.ndarray:
        pass

    @abstractmethod
    def fcd(self, t, xx, yy) -> np.ndarray:
        pass

    @abstractmethod
    def fcs(self, t, xx, yy) -> np.ndarray:
        pass

    def asdict(self) -> Dict[str, Callable]:
        return {"fcp": self.fcp, "fT": self.fT, "fcl": self.fcl, "fcd": self.fcd, "fcs": self.fcs}


class NoForcingTerms(ForcingTermsBase):
    def __init__(self, grid: Grid):
        self._grid = grid

    def _zero(self, t, xx, yy):
        return self._grid.make_full0()

    fcp = fT = fcl = fcd = fcs = _zero


class ForcingTermsFromDict(ForcingTermsBase):
    def __init__(self, forcing_terms_dict: Dict):
        self._forcing_terms_dict = forcing_terms_dict

    def fcp(self, t, xx, yy):
        return self._forcing_terms_dict["fcp"](t, xx, yy)

    def fT(self, t, xx, yy):
        return self._forcing_terms_dict["fT"](t, xx, yy)

    def fcl(self, t, xx, yy):
        return self._forcing_terms_dict["fcl"](t, xx, yy)

    def fcd(self, t, xx, yy):
        return self._forcing_terms_dict["fcd"](t, xx, yy)

    def fcs(self, t, xx, yy):
        return self._forcing_terms_dict["fcs"](t, xx, yy)


# ------------------------------------------------------------------------------------------
# Exact solutions given as plain Python callables, differentiated by second-order finite differences
#                                                          reference src/prob1base.py:892-1155
# ------------------------------------------------------------------------------------------
TimeSteppingStrategy = Literal["forward", "center", "backward"]

# one-dimensional second-order difference formulas as (offset in units of eps, weight) in the order
# the terms are summed; first derivatives are divided by 2*eps, second derivatives by eps*eps
_FD1 = {"center": ((1, 1.0), (-1, -1.0)),
        "forward": ((0, -3.0), (1, 4.0), (2, -1.0)),
        "backward": ((0, 3.0), (-1, -4.0), (-2, 1.0))}
_FD2 = {"center": ((1, 1.0), (0, -2.0), (-1, 1.0)),
        "forward": ((0, 2.0), (1, -5.0), (2, 4.0), (3, -1.0)),
        "backward": ((0, 2.0), (-1, -5.0), (-2, 4.0), (-3, -1.0))}


def pack_analytical_txy_with_o2fdm_derivatives(fn: Callable, *, default_eps: float = 1e-6,
                                               time_stepping: str = "center") -> Callable:
    """fn(t, x, y) -> g(t, x, y, *, d=(dt, dx, dy), op=None, small_eps=None): derivatives of total order
    <= 2 and the Laplacian (op="lap" / "laplacian") by second-order differences of step eps; space
    derivatives are centred, time derivatives follow `time_stepping`
    (reference src/prob1base.py:895-1032)."""
    if time_stepping not in _FD1:
        raise ValueError("Invalid time stepping strategy")

    def combine(terms, at):
        acc = None
        for off, w in terms:
            v = w * at(off)
            acc = v if acc is None else acc + v
        return acc

    def g(t, x, y, *, d=(0, 0, 0), op=None, small_eps=None):
        eps = small_eps or default_eps
        if op is not None:
            if op.lower() not in ("laplacian", "lap"):
                raise ValueError(f"Unknown operator: {op}. Use 'laplacian'/'lap'")
            five = fn(t, x + eps, y) + fn(t, x - eps, y) + fn(t, x, y + eps) + fn(t, x, y - eps) - 4 * fn(t, x, y)
            return five / (eps * eps)
        nt, nx, ny = d
        if any(n not in (0, 1, 2) for n in (nt, nx, ny)):
            raise ValueError("Individual derivatives must be 0, 1, or 2")
        if nt + nx + ny > 2:
            raise ValueError("Combined derivative order must be 0, 1, or 2")
        if nt:   # a time derivative wins over whatever else is asked, as in the reference
            terms = (_FD1 if nt == 1 else _FD2)[time_stepping]
            return combine(terms, lambda k: fn(t + k * eps, x, y)) / ((2 * eps) if nt == 1 else (eps * eps))
        if nx == 1 and ny == 1:
            cross = (fn(t, x + eps, y + eps) - fn(t, x + eps, y - eps) - fn(t, x - eps, y + eps)
                     + fn(t, x - eps, y - eps))
            return cross / (4 * eps * eps)
        if nx:
            terms = (_FD1 if nx == 1 else _FD2)["center"]
            return combine(terms, lambda k: fn(t, x + k * eps, y)) / ((2 * eps) if nx == 1 else (eps * eps))
        if ny:
            terms = (_FD1 if ny == 1 else _FD2)["center"]
            return combine(terms, lambda k: fn(t, x, y + k * eps)) / ((2 * eps) if ny == 1 else (eps * eps))
        return fn(t, x, y)

    return g


class MMSCaseFromAnalytic(MMSCaseBase):
    """Exact solution given as five callables f(t, xx, yy); the derivatives the forcing terms need come
    from pack_analytical_txy_with_o2fdm_derivatives (reference src/prob1base.py:1036-1155).  Such a case
    lives on the host by construction: with the ForcingTerms_* classes its forcing arrays are evaluated
    there and streamed to the device at every time level."""

    def __init__(self, model: DefaultModel01, *, grid: Grid, cp_base, T_base, cl_base, cd_base, cs_base):
        super().__init__(grid, model)
        self.cp_ex = pack_analytical_txy_with_o2fdm_derivatives(cp_base)
        self.T_ex = pack_analytical_txy_with_o2fdm_derivatives(T_base)
        self.cl_ex = pack_analytical_txy_with_o2fdm_derivatives(cl_base)
        self.cd_ex = pack_analytical_txy_with_o2fdm_derivatives(cd_base)
        self.cs_ex = pack_analytical_txy_with_o2fdm_derivatives(cs_base)


def _add_analytic_accessors(cls):
    orders = {"": (0, 0, 0), "dt_": (1, 0, 0), "dx_": (0, 1, 0), "dy_": (0, 0, 1)}
    for var in _VIDX:
        for prefix, d in orders.items():
            def method(self, t, xx, yy, _v=var, _d=d):
                g = getattr(self, _v + "_ex")
                return g(t, xx, yy) if _d == (0, 0, 0) else g(t, xx, yy, d=_d)
            method.__name__ = prefix + var
            setattr(cls, prefix + var, method)
        if var in ("T", "cl", "cd"):
            def lap(self, t, xx, yy, _v=var):
                g = getattr(self, _v + "_ex")
                return g(t, xx, yy, d=(0, 2, 0)) + g(t, xx, yy, d=(0, 0, 2))
            lap.__name__ = "lap_" + var
            setattr(cls, "lap_" + var, lap)
    return cls


_add_analytic_accessors(MMSCaseFromAnalytic)


# ------------------------------------------------------------------------------------------
# MMS cases given as SymPy expressions, and forcing terms evaluated on the host for them
#                                         reference src/prob1base.py:493-598, 1140-1347, 2313-2425
# ------------------------------------------------------------------------------------------
try:  # SymPy is only needed by MMSCaseSymbolic
    import sympy as _sympy
    t_sym, x_sym, y_sym = _sympy.symbols("t x y")
except ImportError:  # pragma: no cover
    _sympy = None
    t_sym = x_sym = y_sym = None


def _lambdify_txy(expr, t_var, x_var, y_var):
    """(t, x, y) -> float64 array of the shape of x; a constant expression is broadcast
    (reference src/prob1base.py:1140-1222).  DiracDelta(a) is 1 where |a| < 1e-13, else 0 (:1257)."""
    mods = [{"DiracDelta": lambda a: np.where(np.abs(a) < 1e-13, 1.0, 0.0)}, "numpy"]
    raw = _sympy.lambdify([t_var, x_var, y_var], expr, modules=mods)

    def fn(t, xx, yy):
        xx, yy = np.asarray(xx, dtype=float), np.asarray(yy, dtype=float)
        assert xx.shape == yy.shape
        val = np.asarray(raw(float(t), xx, yy), dtype=np.float64)
        if val.size == 1:
            return np.full(xx.shape, float(val.reshape(-1)[0]))
        assert val.size == xx.size
        return val.reshape(xx.shape)
    return fn


class MMSCaseSymbolic(MMSCaseBase):
    """Exact solution given as SymPy expressions of (t, x, y); the derivatives the forcing terms need
    are derived symbolically (reference src/prob1base.py:1225-1347).  The accessors cp, dt_cp, ... are
    lambdified for NumPy as in the reference.  For the device the same expressions are lowered to
    register programs (tedd/_symprog.py) that the library runs at every node, so forcing terms, exact
    solution and error norms of such a case never leave the GPU; if an expression uses something the
    program operations do not cover (or device_program=False), the forcing terms are evaluated on the
    host instead and their arrays streamed to the device at every time level."""

    mms_case_id = _ffi.MMS_PROGRAM

    def mms_params(self):
        return ()

    def __init__(self, *, grid: Grid, model: DefaultModel01, cp_sym_expr, T_sym_expr, cl_sym_expr, cd_sym_expr,
                 cs_sym_expr, t_var=None, x_var=None, y_var=None, device_program=True):
        if _sympy is None:
            raise ImportError("MMSCaseSymbolic needs sympy")
        super().__init__(grid, model)
        t_var = t_sym if t_var is None else t_var
        x_var = x_sym if x_var is None else x_var
        y_var = y_sym if y_var is None else y_var
        exprs = {"cp": cp_sym_expr, "T": T_sym_expr, "cl": cl_sym_expr, "cd": cd_sym_expr, "cs": cs_sym_expr}
        named = {}
        for var, expr in exprs.items():
            expr = _sympy.sympify(expr)
            dx, dy = _sympy.diff(expr, x_var), _sympy.diff(expr, y_var)
            pack = {"": expr, "dt_": _sympy.diff(expr, t_var), "dx_": dx, "dy_": dy,
                    "lap_": _sympy.diff(dx, x_var) + _sympy.diff(dy, y_var)}
            for prefix, e in pack.items():
                named[prefix + var] = e
                setattr(self, prefix + var, _lambdify_txy(e, t_var, x_var, y_var))
        self._program = None
        if device_program:
            try:
                self._program = _symprog.lower_case(named, t_var, x_var, y_var)
            except _symprog.Unsupported:
                self._program = None


def avg_int(f: Callable, grid: Grid):
    """Average of f(x, y) over the cells [x_{i-1/2}, x_{i+1/2}] x [y_{j-1/2}, y_{j+1/2}] of the interior
    nodes by 3x3 Gauss-Legendre quadrature; 0 on the boundary (reference src/prob1base.py:493-598)."""
    w = np.array([5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0])
    a = np.array([-np.sqrt(3.0 / 5.0), 0.0, np.sqrt(3.0 / 5.0)])
    x, y = grid.x, grid.y
    x_lo, wx = 0.5 * (x[:-2] + x[1:-1]), grid.h_phalf[1:-1]
    y_lo, wy = 0.5 * (y[:-2] + y[1:-1]), grid.k_phalf[1:-1]
    acc = np.zeros((grid.N - 1, grid.M - 1))
    for i in range(3):
        for j in range(3):
            p, q = np.meshgrid(x_lo + (a[i] + 1.0) * 0.5 * wx, y_lo + (a[j] + 1.0) * 0.5 * wy, indexing="ij")
            acc += w[i] * w[j] * f(p, q)
    out = grid.make_full0()
    out[1:-1, 1:-1] = 0.25 * acc
    return out


class _HostForcingTerms(ForcingTermsBase):
    """Forcing terms of a manufactured solution evaluated with NumPy from the case's accessors
    (cp, dt_cp, dx_cp, ... lap_cd, cs, dt_cs): fcp is the cell average of
    d_t cp + cp (K1 (1 + cl) + K2 T), the others are nodal residuals of the exact solution
    (reference src/prob1base.py:2313-2425; reaction gain g(cs) = cs | H_eta(cs) | (cs > 0):
    :2875-2876, :3503-3551, :3222-3295).  Fields that use them stream the arrays to the device."""

    def __init__(self, mms_case, model, reaction_kind, eta):
        self._mms_case, self._model, self._react, self._eta = mms_case, model, reaction_kind, float(eta)

    @property
    def grid(self):
        return self._mms_case.grid

    def _gain(self, cs):
        if self._react == _ffi.REACT_REGHCSTRIPLE:
            return heaviside_regularized(cs, self._eta)
        if self._react == _ffi.REACT_HCSTRIPLE:
            return (np.asarray(cs) > 0).astype(float)
        return np.asarray(cs, dtype=float)

    def fcp_ptwise(self, t, xx, yy):
        c, m = self._mms_case, self._model
        cp = c.cp(t, xx, yy)
        return c.dt_cp(t, xx, yy) + cp * (m.K1 * (1.0 + c.cl(t, xx, yy)) + m.K2 * c.T(t, xx, yy))

    def fcp(self, t, xx=None, yy=None):
        """Cell averages over the mesh of the coordinates handed in (reference src/prob1base.py:2325-2328:
        avg_int on Grid(xx, yy)); the case's own grid when none are given."""
        grid = self.grid if xx is None or _same_nodes(self.grid, xx, yy) else Grid(xx, yy)
        return avg_int(lambda p, q: self.fcp_ptwise(t, p, q), grid)

    def fT(self, t, xx, yy):
        c, m = self._mms_case, self._model
        return c.dt_T(t, xx, yy) - (m.DT * c.lap_T(t, xx, yy) - m.K3 * c.cp(t, xx, yy) * c.T(t, xx, yy))

    def fcl(self, t, xx, yy):
        c, m = self._mms_case, self._model
        cp, T, cl = c.cp(t, xx, yy), c.T(t, xx, yy), c.cl(t, xx, yy)
        clx, cly = c.dx_cl(t, xx, yy), c.dy_cl(t, xx, yy)
        return c.dt_cl(t, xx, yy) - (
            m.Dl(cp, d=1) * (c.dx_cp(t, xx, yy) * clx + c.dy_cp(t, xx, yy) * cly) + m.Dl(cp) * c.lap_cl(t, xx, yy)
            - m.V1(T) * clx - (cl + 1.0) * m.V1(T, d=1) * c.dx_T(t, xx, yy) - m.K4 * cp * (cl + 1.0))

    def fcd(self, t, xx, yy):
        c, m = self._mms_case, self._model
        cp, T, cd = c.cp(t, xx, yy), c.T(t, xx, yy), c.cd(t, xx, yy)
        dc, dT = m.Dd(cp, T, d=(1, 0)), m.Dd(cp, T, d=(0, 1))
        return c.dt_cd(t, xx, yy) - (
            (dc * c.dx_cp(t, xx, yy) + dT * c.dx_T(t, xx, yy)) * c.dx_cd(t, xx, yy)
            + (dc * c.dy_cp(t, xx, yy) + dT * c.dy_T(t, xx, yy)) * c.dy_cd(t, xx, yy)
            + m.Dd(cp, T) * c.lap_cd(t, xx, yy)
            + m.Kd * (m.Sd - cd) * (c.cl(t, xx, yy) + 1.0) * self._gain(c.cs(t, xx, yy)))

    def fcs(self, t, xx, yy):
        c, m = self._mms_case, self._model
        return c.dt_cs(t, xx, yy) + m.Kd * self._gain(c.cs(t, xx, yy)) * (1.0 + c.cl(t, xx, yy)) * (m.Sd - c.cd(t, xx, yy))


class _DeviceForcingTerms(ForcingTermsBase):
    """MMS forcing terms evaluated by the device library from a closed-form MMS case:
    cell-averaged fcp (3x3 Gauss-Legendre, src/prob1base.py:493-598, 2313-2328) and nodal
    fT, fcl, fcd, fcs (src/prob1base.py:2331-2425, 3503-3551)."""

    reaction_kind = None

    def __init__(self, *, mms_case: MMSCaseBase, model: DefaultModel01, eta=0.0):
        # a case the library has no closed form for: evaluate on the host, the field streams the arrays
        on_device = isinstance(mms_case, MMSCaseDevice) or getattr(mms_case, "_program", None) is not None
        self._host = None if on_device else _HostForcingTerms(mms_case, model, self.reaction_kind, eta)
        self._mms_case = mms_case
        self._model = model
        self._eta = float(eta)
        self._session = None
        self._meshes = _MeshSessions()
        self._memo = {}

    @property
    def grid(self):
        return self._mms_case.grid

    @property
    def mms_case(self):
        return self._mms_case

    @property
    def model(self):
        return self._model

    def _eval(self, t, xx, yy):
        if self._host is not None:
            g = self.grid
            xx, yy = (g.xx, g.yy) if xx is None else (xx, yy)
            key = ("host", float(t))
            if key not in self._memo or not _same_nodes(g, xx, yy):
                vals = [self._host.fcp(t, xx, yy), self._host.fT(t, xx, yy), self._host.fcl(t, xx, yy),
                        self._host.fcd(t, xx, yy), self._host.fcs(t, xx, yy)]
                if not _same_nodes(g, xx, yy):
                    return vals
                self._memo.clear()
                self._memo[key] = vals
            return self._memo[key]
        grid = self._meshes.grid_for(self.grid, xx, yy)
        s = self._meshes.session(self.grid, grid, self._session, self._make_session)
        if grid is self.grid:
            self._session = s
        s.refresh(self._model, self._eta, self._mms_case.mms_params())
        key = (float(t), s._sent, id(grid))
        if key not in self._memo:
            self._memo.clear()
            self._memo[key] = s.ctx.eval_forcing(t)
            self._memo["grid"] = grid      # keeps id(grid) unambiguous while the entry lives
        return self._memo[key]

    def _make_session(self, grid):
        return _Session(grid, self._model, self.reaction_kind, self._eta, self._mms_case.mms_case_id,
                        self._mms_case.mms_params(), program=getattr(self._mms_case, "_program", None))

    def fcp(self, t, xx=None, yy=None):
        return self._eval(t, xx, yy)[0].copy()

    def fT(self, t, xx=None, yy=None):
        return self._eval(t, xx, yy)[1].copy()

    def fcl(self, t, xx=None, yy=None):
        return self._eval(t, xx, yy)[2].copy()

    def fcd(self, t, xx=None, yy=None):
        return self._eval(t, xx, yy)[3].copy()

    def fcs(self, t, xx=None, yy=None):
        return self._eval(t, xx, yy)[4].copy()


class ForcingTerms_CsTriple(_DeviceForcingTerms):
    reaction_kind = _ffi.REACT_CSTRIPLE

    def __init__(self, *, mms_case: MMSCaseBase, model: DefaultModel01):
        super().__init__(mms_case=mms_case, model=model, eta=0.0)


class ForcingTerms_HCsTriple(_DeviceForcingTerms):
    """Forcing with the sharp Heaviside H(cs) = (cs > 0) in fcd, fcs (reference src/prob1base.py:3222-3295)."""

    reaction_kind = _ffi.REACT_HCSTRIPLE

    def __init__(self, *, mms_case: MMSCaseBase, model: DefaultModel01):
        super().__init__(mms_case=mms_case, model=model, eta=0.0)


class ForcingTerms_RegHCsTriple(_DeviceForcingTerms):
    reaction_kind = _ffi.REACT_REGHCSTRIPLE

    def __init__(self, *, mms_case: MMSCaseBase, model: DefaultModel01, regularization_factor: float):
        super().__init__(mms_case=mms_case, model=model, eta=regularization_factor)
        self._regularization_factor = regularization_factor

    @property
    def regularization_factor(self):
        return self._regularization_factor


def heaviside_regularized(x: np.ndarray, regularization_factor: float) -> np.ndarray:
    """H_eta(x) = 1 / (1 + exp(-eta x))   (reference src/prob1base.py:3452-3466)."""
    with np.errstate(over="ignore"):
        return 1 / (1 + np.exp(-regularization_factor * np.asarray(x, dtype=float)))


def state_from_mms_when(*, mms_case, t, grid):
    """StateVars holding the exact MMS solution at time t (reference src/prob1base.py:3433-3449)."""
    if isinstance(mms_case, MMSCaseDevice) and mms_case.grid is grid:
        cp, T, cl, cd, cs = mms_case.exact_fields(t)
    else:
        xx, yy = grid.xx, grid.yy
        cp, T, cl, cd, cs = (getattr(mms_case, v)(t, xx, yy) for v in ("cp", "T", "cl", "cd", "cs"))
    return StateVars(cp, T, cl, cd, cs, model=mms_case.model, hh=grid.hh, kk=grid.kk)


# ------------------------------------------------------------------------------------------
# semi-discrete fields                              reference src/prob1base.py:2133-2293, 2429-2876, 3553-3593
# ------------------------------------------------------------------------------------------
_OFFSET_INDEX = {(0, 0): 0, (1, 0): 1, (-1, 0): 2, (0, 1): 3, (0, -1): 4}


class SemiDiscreteFieldBase(ABC):
    def __init__(self, *, grid: Grid, model: DefaultModel01):
        self._model = model
        self._grid = grid

    @property
    def model(self):
        return self._model

    @property
    def grid(self):
        return self._grid


class _DeviceField(SemiDiscreteFieldBase):
    """F = (Fcp, FT, Fcl, Fcd, Fcs) and the Jacobian stencils, evaluated on the device.

    The forcing callables fcp, fT, fcl, fcd, fcs are copied from `forcing_terms` onto the field
    (as the reference does, src/prob1base.py:2464-2467) and may be re-assigned by the caller.
    If they still are the device forcing terms of a closed-form MMS case with this field's
    reaction kind, the kernels evaluate the forcing in place; otherwise the five arrays are
    obtained from the callables at the needed times and streamed to the device."""

    reaction_kind = None
    _forcing_names = ("fcp", "fT", "fcl", "fcd", "fcs")

    def __init__(self, *, grid, model, forcing_terms: ForcingTermsBase, eta=0.0):
        super().__init__(grid=grid, model=model)
        self._forcing_terms = forcing_terms
        for name in self._forcing_names:
            setattr(self, name, getattr(forcing_terms, name))
        self._eta = eta
        self._sessions = {}
        self._field_memo = None

    def _eta_now(self):
        return float(self._eta)

    def _closed_form_source(self):
        """The device forcing terms if every forcing callable still is its bound method."""
        ft = self._forcing_terms
        if not isinstance(ft, _DeviceForcingTerms) or ft.reaction_kind != self.reaction_kind:
            return None
        if ft._host is not None:  # no closed form in the library: host-evaluated, streamed
            return None
        # (the grid the case was built with does not matter: a closed form is a function of the coordinates,
        # and the reference hands the field's own node coordinates to the forcing callables)
        if ft.model is not self._model:
            return None
        if self.reaction_kind == _ffi.REACT_REGHCSTRIPLE and float(ft._eta) != self._eta_now():
            return None
        for name in self._forcing_names:
            fn = getattr(self, name)
            if getattr(fn, "__self__", None) is not ft or getattr(fn, "__func__", None) is not getattr(type(ft), name):
                return None
        return ft

    def _device(self):
        src = self._closed_form_source()
        if src is not None:
            case = src.mms_case
            key = ("mms", case.mms_case_id, id(case))
            mms_id, params, program = case.mms_case_id, case.mms_params(), getattr(case, "_program", None)
        else:
            key = ("stream",)
            mms_id, params, program = _ffi.MMS_STREAM, (), None
        s = self._sessions.get(key)
        if s is None:
            s = _Session(self._grid, self._model, self.reaction_kind, self._eta_now(), mms_id, params, program=program)
            self._sessions = {key: s}
        s.refresh(self._model, self._eta_now(), params)
        return s

    def _stream_forcing(self, session, which, t):
        g = self._grid
        arrs = [np.asarray(getattr(self, n)(t, g.xx, g.yy), dtype=float) for n in self._forcing_names]
        session.ctx.set_forcing(which, *arrs)

    def _eval_F(self, at_t: StateVars, t):
        s = self._device()
        memo = self._field_memo
        if memo is not None and memo[0] is at_t and memo[1] == float(t) and memo[2] == s._sent and s.mms_case != _ffi.MMS_STREAM:
            return memo[3]
        s.load_state(at_t)
        if s.mms_case == _ffi.MMS_STREAM:
            self._stream_forcing(s, 0, t)
        out = s.ctx.eval_field(t)
        self._field_memo = (at_t, float(t), s._sent, out)
        return out

    def Fcp(self, at_t: StateVars, t: float):
        return self._eval_F(at_t, t)[0].copy()

    def FT(self, at_t: StateVars, t: float):
        return self._eval_F(at_t, t)[1].copy()

    def Fcl(self, at_t: StateVars, t: float):
        return self._eval_F(at_t, t)[2].copy()

    def Fcd(self, at_t: StateVars, t: float):
        return self._eval_F(at_t, t)[3].copy()

    def Fcs(self, at_t: StateVars, t: float):
        return self._eval_F(at_t, t)[4].copy()

    def _jac(self, op, state, a, b):
        assert a in (-1, 0, 1) and b in (-1, 0, 1) and (a == 0 or b == 0)
        s = self._device()
        s.load_state(state)
        return s.ctx.eval_jacobian(op)[_OFFSET_INDEX[(a, b)]]

    def delT_ab_FT_ij(self, state, *, a, b):
        return self._jac(0, state, a, b)

    def delT_ab_Fcl_ij(self, state, *, a, b):
        return self._jac(1, state, a, b)

    def delcl_ab_Fcl_ij(self, state, *, a, b):
        return self._jac(2, state, a, b)

    def delT_ab_Fcd_ij(self, state, *, a, b):
        return self._jac(3, state, a, b)

    def delcl_ab_Fcd_ij(self, state, *, a, b):
        return self._jac(4, state, a, b)

    def delcd_ab_Fcd_ij(self, state, *, a, b):
        return self._jac(5, state, a, b)

    # ---- the Cs/Cd interaction law, stated for callers that inspect it ------------------------------
    # [Cs-Cd-int] = F1(cp) (a_T T + b_T) (a_cl cl + b_cl) (a_cd cd + b_cd) F2(cs) with null boundary values
    # (reference src/prob1base.py:2440-2597; the three families: :2863-2876, :3320-3337, :3580-3593).  These
    # NumPy accessors describe the law the device kernels have built in for the field's reaction kind; the
    # kernels do not call them, so overriding one in a subclass does not change a step.
    def cscd_reaction_T(self):
        return (0, 1)

    def cscd_reaction_cl(self):
        return (1, 1)

    def cscd_reaction_cd(self):
        return (-1, self.model.Sd)

    def cscd_reaction_cp(self, cp):
        return self.grid.const_with_nullbd(1)

    def cscd_reaction_cs(self, cs):
        cs = np.asarray(cs, dtype=float)
        if self.reaction_kind == _ffi.REACT_REGHCSTRIPLE:
            return self.model.Kd * heaviside_regularized(cs, self._eta_now())
        if self.reaction_kind == _ffi.REACT_HCSTRIPLE:
            return self.model.Kd * (cs > 0)
        return self.model.Kd * cs

    def _cscd_product(self, state, slope_of=None):
        """The interaction term, or its derivative with respect to the nodal value of `slope_of`."""
        out = self.cscd_reaction_cp(state.cp) * self.cscd_reaction_cs(state.cs)
        for name, (slope, offset) in (("T", self.cscd_reaction_T()), ("cl", self.cscd_reaction_cl()),
                                      ("cd", self.cscd_reaction_cd())):
            if name == slope_of:
                if slope == 0.0:
                    return self.grid.make_full0()
                out = out * slope
            else:
                out = out * (slope * getattr(state, name) + offset)
        return out * self.grid.null_bd_mask

    def cscd_reaction_term(self, state: StateVars):
        return self._cscd_product(state)

    def _del_cscd(self, state, name, a, b):
        assert a in (-1, 0, 1) and b in (-1, 0, 1) and (a == 0 or b == 0)
        return self._cscd_product(state, slope_of=name) if a == 0 and b == 0 else self.grid.make_full0()

    def delT_ab_cscd_reaction_ij(self, state: StateVars, *, a, b):
        return self._del_cscd(state, "T", a, b)

    def delcl_ab_cscd_reaction_ij(self, state: StateVars, *, a, b):
        return self._del_cscd(state, "cl", a, b)

    def delcd_ab_cscd_reaction_ij(self, state: StateVars, *, a, b):
        return self._del_cscd(state, "cd", a, b)


class SemiDiscreteField_CsTriple(_DeviceField):
    """[Cs-Cd-int] = Kd (Sd - Cd)(1 + Cl) Cs   (reference src/prob1base.py:2842-2876)."""

    reaction_kind = _ffi.REACT_CSTRIPLE

    def __init__(self, *, grid: Grid, model: DefaultModel01, forcing_terms: ForcingTermsBase):
        super().__init__(grid=grid, model=model, forcing_terms=forcing_terms, eta=0.0)


class SemiDiscreteField_HCsTriple(_DeviceField):
    """[Cs-Cd-int] = Kd (Sd - Cd)(1 + Cl) H(Cs), H(cs) = (cs > 0)   (reference src/prob1base.py:3298-3337)."""

    reaction_kind = _ffi.REACT_HCSTRIPLE

    def __init__(self, *, grid: Grid, model: DefaultModel01, forcing_terms: ForcingTermsBase):
        super().__init__(grid=grid, model=model, forcing_terms=forcing_terms, eta=0.0)


class SemiDiscreteField_RegHCsTriple(_DeviceField):
    """[Cs-Cd-int] = Kd (Sd - Cd)(1 + Cl) H_eta(Cs)   (reference src/prob1base.py:3553-3593)."""

    reaction_kind = _ffi.REACT_REGHCSTRIPLE

    def __init__(self, *, grid: Grid, model: DefaultModel01, forcing_terms: ForcingTermsBase,
                 regularization_factor: float):
        super().__init__(grid=grid, model=model, forcing_terms=forcing_terms, eta=regularization_factor)
        self._regularization_factor = regularization_factor

    @property
    def regularization_factor(self):
        return self._regularization_factor

    def _eta_now(self):
        return float(self._regularization_factor)


# ------------------------------------------------------------------------------------------
# time integrators                         reference src/prob1base.py:2879-3219, 3596-3702
# ------------------------------------------------------------------------------------------
class TimeIntegratorBase(ABC):
    @abstractmethod
    def step(self, at_t0: StateVars, *, t0, dt):
        pass


# Methods of a field whose work the step kernels do themselves: the reference's integrators call them on the
# field object (src/prob1base.py:2885-3149), so a test or a user can change a step by overriding one
# (tests/test_time_integrator_hcs_triple.py:208-226 overrides Fcs).  The kernels do not call back into Python:
# an override would be ignored, so it is refused.  (Re-assigned forcing callables f* are honoured: streamed.)
_KERNEL_METHODS = (
    "Fcp", "FT", "Fcl", "Fcd", "Fcs",
    "delT_ab_FT_ij", "delT_ab_Fcl_ij", "delcl_ab_Fcl_ij", "delT_ab_Fcd_ij", "delcl_ab_Fcd_ij", "delcd_ab_Fcd_ij",
    "delT_Fcl_at_w", "delT_Fcd_at_w", "delcl_Fcd_at_w",
    "cscd_reaction_T", "cscd_reaction_cl", "cscd_reaction_cd", "cscd_reaction_cp", "cscd_reaction_cs",
    "cscd_reaction_term", "delT_ab_cscd_reaction_ij", "delcl_ab_cscd_reaction_ij", "delcd_ab_cscd_reaction_ij",
)


def _overridden_kernel_methods(field):
    """Names of kernel-implemented methods that `field` (its instance dictionary or a class outside this module)
    replaces."""
    found = []
    for name in _KERNEL_METHODS:
        if name in vars(field):
            found.append(name)
            continue
        for klass in type(field).__mro__:
            if name in vars(klass):
                if klass.__module__ != __name__:
                    found.append(name)
                break
    return found


def _require_device_field(field):
    if not isinstance(field, _DeviceField):
        raise NotImplementedError(
            f"{type(field).__name__} is not a device field: the integrators of this package run on the GPU and "
            "have no CPU fallback")
    replaced = _overridden_kernel_methods(field)
    if replaced:
        raise NotImplementedError(
            f"{type(field).__name__} overrides {', '.join(replaced)}: the step runs in device kernels that have the "
            "field, its Jacobian stencils and the interaction laws built in and do not call back into Python, so the "
            "override would be ignored. Express the change through the forcing callables f* (they may be re-assigned "
            "and are then streamed to the device) or through the model constants.")
    return field


class _LazyResiduals(Mapping):
    """`last_residual` of the predictor-corrector integrators: the Newton residuals of T, cl, cd
    of the last step, fetched from the device on first access."""

    _keys = ("T", "cl", "cd")

    def __init__(self, fetch=None):
        self._fetch = fetch
        self._data = None

    def _load(self):
        if self._data is None:
            self._data = {} if self._fetch is None else dict(zip(self._keys, self._fetch()))
        return self._data

    def __getitem__(self, k):
        return self._load()[k]

    def __iter__(self):
        return iter(self._load())

    def __len__(self):
        return len(self._load())


class ForwardEulerIntegrator(TimeIntegratorBase):
    """X1 = X0 + dt F(X0, t0) on the full arrays (reference src/prob1base.py:2885-2903)."""

    def __init__(self, semi_discrete_field: SemiDiscreteFieldBase, **kwargs):
        self.semi_discrete_field = semi_discrete_field

    def step(self, at_t0: StateVars, *, t0, dt):
        field = _require_device_field(self.semi_discrete_field)
        s = field._device()
        s.load_state(at_t0)
        if s.mms_case == _ffi.MMS_STREAM:
            field._stream_forcing(s, 0, t0)
        s.ctx.step_feuler(t0, dt)
        cp, T, cl, cd, cs = s.ctx.get_state()
        out = at_t0.with_changes(cp=cp, T=T, cl=cl, cd=cd, cs=cs)
        s.state_token = out
        return out


class P_ModifiedEuler_C_Trapezoidal_TimeIntegratorBase(ABC):
    """Predictor-corrector step: Heun predictor for cp, cs; `num_newton_steps` Newton steps of the
    trapezoidal rule for T -> cl -> cd; trapezoidal corrector for cp, cs; repeated `num_pc_steps`
    times (reference src/prob1base.py:2906-3149).  The whole step runs on the device; the linear
    systems are solved by matrix-free Jacobi sweeps instead of spsolve.

    Extra keyword arguments of this implementation
      jacobi_tol     target for the last sweep's update relative to the solution increment (1e-13)
      jacobi_max     cap on the sweeps per system
      jacobi_sweeps  fixed sweep counts (T, cl, cd), overriding the automatic choice
    """

    _reaction_kind = None

    def __init__(self, semi_discrete_field, *, num_pc_steps=1, num_newton_steps=1,
                 jacobi_tol=1e-13, jacobi_max=100000, jacobi_sweeps=None):
        self.semi_discrete_field = semi_discrete_field
        self._model = semi_discrete_field.model
        self._grid = semi_discrete_field.grid
        self.num_pc_steps = num_pc_steps
        self.num_newton_steps = num_newton_steps
        self.jacobi_tol = jacobi_tol
        self.jacobi_max = jacobi_max
        self.jacobi_sweeps = jacobi_sweeps
        self.last_residual: Mapping = _LazyResiduals()
        self.last_stats: Dict = {}

    def _cs_newton_args(self):
        return dict(cs_newton_iterations=0, cs_consec_xs_rtol=0.0)

    def _pc_kwargs(self, sweeps=None):
        kw = dict(num_pc_steps=self.num_pc_steps, num_newton_steps=self.num_newton_steps,
                  jacobi_sweeps=sweeps if sweeps is not None else self.jacobi_sweeps,
                  jacobi_tol=self.jacobi_tol, jacobi_max=self.jacobi_max)
        kw.update(self._cs_newton_args())
        return kw

    def _session(self):
        field = _require_device_field(self.semi_discrete_field)
        if field.reaction_kind != self._reaction_kind:
            raise NotImplementedError(f"{type(self).__name__} needs a field with the matching reaction term")
        if field.model is not self._model:
            # tests swap the model on the integrator; the field's model is what the kernels use
            field._model = self._model
        return field, field._device()

    # ---- the pieces of step(), callable on their own like the reference's (src/prob1base.py:2946-3115)
    def _substep_session(self, state, t0, dt):
        field, s = self._session()
        s.load_state(state)
        if s.mms_case == _ffi.MMS_STREAM:
            field._stream_forcing(s, 0, t0)
            field._stream_forcing(s, 1, t0 + dt)
        return s

    def initial_cp_pred(self, at_t, t, *, dt):
        """Heun predictor for cp with the other fields frozen (reference src/prob1base.py:2953-2965)."""
        return self._substep_session(at_t, t, dt).ctx.predict_cp_cs(t, dt)[0]

    def initial_cs_pred(self, at_t, t, *, dt):
        """Heun predictor for cs (reference src/prob1base.py:3174-3188, 3354-3374, 3631-3645)."""
        return self._substep_session(at_t, t, dt).ctx.predict_cp_cs(t, dt)[1]

    def corrector_cp_step(self, T1, cl1, _cd1_ignroed=None, *, at_t0, t0, dt, _cd1_ignored=None):
        """Closed-form trapezoidal corrector for cp (reference src/prob1base.py:2967-2996).  The third
        parameter keeps the reference's spelling, which its tests pass by keyword
        (tests/test_isolated_correctors_cp_cs.py:107); the corrected spelling is accepted too."""
        assert dt > 0
        s = self._substep_session(at_t0, t0, dt)
        cd1 = _cd1_ignroed if _cd1_ignroed is not None else _cd1_ignored
        cd1 = cd1 if cd1 is not None else at_t0.cd
        return s.ctx.correct_cp_cs(t0, dt, T1, cl1, cd1, **self._cs_newton_args())[0]

    def corrector_cs_step(self, _T1_ignored, cl1, cd1, *, at_t0, t0, dt):
        """Corrector for cs (reference src/prob1base.py:3190-3219, 3376-3430, 3665-3702)."""
        assert dt > 0
        s = self._substep_session(at_t0, t0, dt)
        T1 = _T1_ignored if _T1_ignored is not None else at_t0.T
        return s.ctx.correct_cp_cs(t0, dt, T1, cl1, cd1, **self._cs_newton_args())[1]

    def _newton(self, var, name, at_t0, t0, dt, Y, T1=None, cl1=None):
        s = self._substep_session(at_t0, t0, dt)
        sweeps = 0 if self.jacobi_sweeps is None else int(self.jacobi_sweeps[var - 1])
        out, res = s.ctx.newton_step(var, t0, dt, Y, T1, cl1, jacobi_sweeps=sweeps, jacobi_tol=self.jacobi_tol,
                                     jacobi_max=self.jacobi_max)
        if not isinstance(self.last_residual, dict):
            self.last_residual = {}
        self.last_residual[name] = res
        return out

    def newton_step_T(self, at_t0: StateVars, *, t0, dt, YT0):
        """One Newton step of the trapezoidal rule for T (reference src/prob1base.py:2998-3045)."""
        return self._newton(1, "T", at_t0, t0, dt, YT0)

    def newton_step_cl(self, at_t0, T1, *, t0, dt, Ycl0):
        """... for cl given the new T (reference src/prob1base.py:3047-3080)."""
        return self._newton(2, "cl", at_t0, t0, dt, Ycl0, T1)

    def newton_step_cd(self, at_t0, T1, cl1, *, t0, dt, Ycd0):
        """... for cd given the new T, cl (reference src/prob1base.py:3082-3115)."""
        return self._newton(3, "cd", at_t0, t0, dt, Ycd0, T1, cl1)

    def step(self, at_t0: StateVars, *, t0, dt):
        assert dt > 0
        field, s = self._session()
        s.load_state(at_t0)
        if s.mms_case == _ffi.MMS_STREAM:
            field._stream_forcing(s, 0, t0)
            field._stream_forcing(s, 1, t0 + dt)
        sweeps = None
        for _attempt in range(12):
            s.ctx.step_pc(t0, dt, **self._pc_kwargs(sweeps))
            ok = True
            if self.num_pc_steps > 0 and self.num_newton_steps > 0 and self.jacobi_sweeps is None:
                if sweeps is None:
                    # sweep counts the library chose are judged by the library (tedd_check_converged)
                    try:
                        s.ctx.check_converged()
                    except RuntimeError:
                        ok = False
                    stats = s.ctx.get_stats()
                else:
                    # counts fixed here (the redo below) by the record of the step
                    stats = s.ctx.get_stats()
                    for v in range(3):
                        upd, inc = stats["last_update"][v], stats["increment"][v]
                        if not (upd <= 10 * self.jacobi_tol * inc or upd == 0.0):
                            ok = False
            else:
                stats = s.ctx.get_stats()
            if ok:
                break
            # the a-priori contraction bound was too optimistic for this state: redo the step
            # from the same input with twice the sweeps
            sweeps = [min(self.jacobi_max, 2 * n) for n in stats["sweeps"]]
            s.state_token = None
            s.load_state(at_t0)
            if s.mms_case == _ffi.MMS_STREAM:
                field._stream_forcing(s, 0, t0)
                field._stream_forcing(s, 1, t0 + dt)
        else:
            raise RuntimeError("Jacobi sweeps did not reach jacobi_tol")
        self.last_stats = stats
        cp, T, cl, cd, cs = s.ctx.get_state()
        out = at_t0.with_changes(cp=cp, T=T, cl=cl, cd=cd, cs=cs)
        s.state_token = out
        if self.num_pc_steps > 0 and self.num_newton_steps > 0:
            ctx = s.ctx
            self.last_residual = _LazyResiduals(lambda: ctx.get_residuals())
        return out

    # -- fused multi-step run with the error norms of mms_trial_utils.run_simulation_collect_data
    def advance_with_errors(self, initial_state: StateVars, *, t0, dt, num_steps):
        """Runs num_steps steps on the device and returns (final StateVars, err_series) where
        err_series[k] = [||e_cp||_H^2, ||e_T||_H^2, ||e_cl||_H^2, ||e_cd||_H^2, ||e_cs||_H^2,
        ||grad e_T||_p^2, ||grad e_cl||_p^2, ||grad e_cd||_p^2] at t0 + k dt.  Only for closed-form
        MMS forcing."""
        field, s = self._session()
        if s.mms_case == _ffi.MMS_STREAM:
            raise NotImplementedError("fused error norms need a closed-form MMS case")
        s.load_state(initial_state)
        series = s.ctx.advance(t0, dt, num_steps, integrator=1, **self._pc_kwargs())
        self.last_stats = s.ctx.get_stats()
        cp, T, cl, cd, cs = s.ctx.get_state()
        out = initial_state.with_changes(cp=cp, T=T, cl=cl, cd=cd, cs=cs)
        s.state_token = out
        ctx = s.ctx
        self.last_residual = _LazyResiduals(lambda: ctx.get_residuals())
        return out, series


class P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_CsTriple(P_ModifiedEuler_C_Trapezoidal_TimeIntegratorBase):
    """CsTriple variant: closed-form trapezoidal corrector for cs (reference src/prob1base.py:3152-3219)."""

    _reaction_kind = _ffi.REACT_CSTRIPLE

    def __init__(self, semi_discrete_field, *, num_pc_steps=1, num_newton_steps=1, **device_kwargs):
        super().__init__(semi_discrete_field, num_pc_steps=num_pc_steps, num_newton_steps=num_newton_steps,
                         **device_kwargs)


class P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_HCsTriple(P_ModifiedEuler_C_Trapezoidal_TimeIntegratorBase):
    """HCsTriple variant: Heun predictor for cs times the interior mask; cs corrector by the sign of
    Y0 = 2 cs0 + dt Fcs(S0, t0) + dt fcs1 (cs1 = Y0 / (2 - dt R1) if Y0 > 0, Y0 / 2 if Y0 < 0, else 0),
    raising ValueError when 2 - dt Kd (Sd - cd1)(1 + cl1) is not positive  (reference
    src/prob1base.py:3340-3430)."""

    _reaction_kind = _ffi.REACT_HCSTRIPLE

    def __init__(self, semi_discrete_field, *, num_pc_steps=1, num_newton_steps=1, **device_kwargs):
        super().__init__(semi_discrete_field, num_pc_steps=num_pc_steps, num_newton_steps=num_newton_steps,
                         **device_kwargs)


class P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_RegHCsTriple(P_ModifiedEuler_C_Trapezoidal_TimeIntegratorBase):
    """RegHCsTriple variant: cs corrector by per-node scalar Newton on
    2x + (2x - y) e^{-eta x} = y - a  (reference src/prob1base.py:3596-3702)."""

    _reaction_kind = _ffi.REACT_REGHCSTRIPLE

    def __init__(self, semi_discrete_field, *, num_pc_steps=1, num_newton_steps=1,
                 regularization_factor: float, num_newton_iterations: int = 5, consec_xs_rtol: float = 1e-6,
                 **device_kwargs):
        super().__init__(semi_discrete_field, num_pc_steps=num_pc_steps, num_newton_steps=num_newton_steps,
                         **device_kwargs)
        self._regularization_factor = regularization_factor
        self._num_newton_iterations = num_newton_iterations
        self._consec_xs_rtol = consec_xs_rtol

    def _cs_newton_args(self):
        return dict(cs_newton_iterations=int(self._num_newton_iterations),
                    cs_consec_xs_rtol=float(self._consec_xs_rtol))
