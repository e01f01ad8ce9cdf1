"""SymPy expressions -> straight-line register programs for the device library.

The reference lambdifies the expressions of an MMSCaseSymbolic for NumPy (src/prob1base.py:1225-1347);
here they are lowered, after common subexpression elimination, to three-address operations on double
registers (word format and operation codes: include/tedd.h, tedd_set_mms_program) that every node runs
on the GPU.  `lower` raises Unsupported for anything the operation set does not cover (Piecewise,
special functions, ...): the caller then keeps evaluating that case on the host.

`run_numpy` interprets a program with NumPy arrays; the tests use it to check the lowering against
sympy.lambdify without a GPU.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

try:
    import sympy
except ImportError:  # pragma: no cover
    sympy = None

(OP_CONST, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_NEG, OP_POWI, OP_POW, OP_EXP, OP_SIN, OP_COS, OP_SQRT, OP_LOG,
 OP_ABS, OP_SIGN, OP_TANH, OP_OUT, OP_MOV) = range(18)
MAX_REGS = 96
_BINARY = (OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_POW)


class Unsupported(Exception):
    pass


class _Emitter:
    """Emits operations on virtual registers (single assignment); (op, dst, a, b) with b an int payload
    for CONST (constant index) and POWI (exponent)."""

    def __init__(self, t, x, y):
        self.ops: List[Tuple[int, int, int, int]] = []
        self.consts: List[float] = []
        self.const_index = {}
        self.sym = {t: 0, x: 1, y: 2}
        self.next = 3
        self._size = {}

    def size(self, e):
        """Number of nodes of the expression tree: operands are evaluated largest first, so that the
        value of a small operand (a constant, a symbol) is not kept in a register across a big one."""
        n = self._size.get(e)
        if n is None:
            n = 1 + sum(self.size(a) for a in e.args)
            self._size[e] = n
        return n

    def new(self):
        self.next += 1
        return self.next - 1

    def op(self, code, a, b=0):
        d = self.new()
        self.ops.append((code, d, a, b))
        return d

    def const(self, value):
        value = float(value)
        key = value.hex()
        if key not in self.const_index:
            self.const_index[key] = len(self.consts)
            self.consts.append(value)
        if len(self.consts) > 65536:
            raise Unsupported("too many distinct constants")
        return self.op(OP_CONST, 0, self.const_index[key])

    def emit(self, e):
        S = sympy
        if e.is_Symbol:
            if e not in self.sym:
                raise Unsupported(f"free symbol {e}")
            return self.sym[e]
        if e.is_number:
            if not e.is_real and e.is_real is not None:
                raise Unsupported(f"non-real constant {e}")
            return self.const(S.N(e, 30))
        if e.is_Add:
            pos, neg = [], []
            for a in sorted(e.args, key=self.size, reverse=True):
                (neg if a.could_extract_minus_sign() else pos).append(a)
            if pos:
                acc = self.emit(pos[0])
                for a in pos[1:]:
                    acc = self.op(OP_ADD, acc, self.emit(a))
                for a in neg:
                    acc = self.op(OP_SUB, acc, self.emit(-a))
            else:
                acc = self.op(OP_NEG, self.emit(-neg[0]))
                for a in neg[1:]:
                    acc = self.op(OP_SUB, acc, self.emit(-a))
            return acc
        if e.is_Mul:
            coeff, rest = e.as_coeff_Mul()
            negate = False
            if coeff.is_number and coeff.could_extract_minus_sign():
                negate, coeff = True, -coeff
            num, den = [], []
            for a in sorted(S.Mul.make_args(rest), key=self.size, reverse=True):
                if a.is_Pow and a.exp.is_number and a.exp.could_extract_minus_sign():
                    den.append(S.Pow(a.base, -a.exp))
                else:
                    num.append(a)
            if coeff != 1:
                num.append(coeff)
            acc = self.emit(num[0]) if num else self.const(1.0)
            for a in num[1:]:
                acc = self.op(OP_MUL, acc, self.emit(a))
            for a in den:
                acc = self.op(OP_DIV, acc, self.emit(a))
            return self.op(OP_NEG, acc) if negate else acc
        if e.is_Pow:
            base, ex = e.base, e.exp
            if base == S.E:
                return self.op(OP_EXP, self.emit(ex))
            if ex.is_Integer and -127 <= int(ex) <= 127:
                return self.op(OP_POWI, self.emit(base), int(ex) & 0xff)
            if ex == S.Rational(1, 2):
                return self.op(OP_SQRT, self.emit(base))
            if ex == S.Rational(-1, 2):
                return self.op(OP_DIV, self.const(1.0), self.op(OP_SQRT, self.emit(base)))
            return self.op(OP_POW, self.emit(base), self.emit(ex))
        unary = {S.exp: OP_EXP, S.sin: OP_SIN, S.cos: OP_COS, S.log: OP_LOG, S.Abs: OP_ABS, S.sign: OP_SIGN,
                 S.tanh: OP_TANH}
        if e.func in unary and len(e.args) == 1:
            return self.op(unary[e.func], self.emit(e.args[0]))
        if e.func == S.tan:
            a = self.emit(e.args[0])
            return self.op(OP_DIV, self.op(OP_SIN, a), self.op(OP_COS, a))
        if e.func in (S.cosh, S.sinh):
            a = self.emit(e.args[0])
            ep, em = self.op(OP_EXP, a), self.op(OP_EXP, self.op(OP_NEG, a))
            s = self.op(OP_ADD if e.func == S.cosh else OP_SUB, ep, em)
            return self.op(OP_MUL, s, self.const(0.5))
        raise Unsupported(f"no device operation for {e.func}")


def _allocate(ops, outputs, max_regs):
    """Virtual registers -> at most max_regs physical ones (registers 0..2 are t, x, y and stay)."""
    last_use = {}
    for pos, (code, d, a, b) in enumerate(ops):
        if code != OP_CONST:
            last_use[a] = pos
            if code in _BINARY:
                last_use[b] = pos
    for slot, v in outputs:
        last_use[v] = len(ops) + slot
    phys = {0: 0, 1: 1, 2: 2}
    free = []
    n_regs = 3
    out = []
    for pos, (code, d, a, b) in enumerate(ops):
        pa = phys[a] if code != OP_CONST else 0
        pb = phys[b] if code in _BINARY else b
        # operands that die here can be reused for the destination
        for v in ((a,) if code != OP_CONST else ()) + ((b,) if code in _BINARY else ()):
            if v > 2 and last_use.get(v) == pos and v in phys:
                free.append(phys.pop(v))
        if d not in last_use:  # dead value
            continue
        if free:
            pd = free.pop()
        else:
            pd = n_regs
            n_regs += 1
            if n_regs > max_regs:
                raise Unsupported(f"needs more than {max_regs} registers")
        phys[d] = pd
        out.append((code, pd, pa, pb))
    for slot, v in outputs:
        out.append((OP_OUT, slot, phys[v], 0))
    return out, n_regs


def lower(exprs: Sequence, t, x, y, max_regs: int = MAX_REGS):
    """(code uint32[n], consts float64[m], n_regs) computing out[k] = exprs[k](t, x, y)."""
    if sympy is None:
        raise Unsupported("sympy is not importable")
    exprs = [sympy.sympify(e) for e in exprs]
    em = _Emitter(t, x, y)
    try:
        repl, reduced = sympy.cse(exprs, order="none")
        for s, e in repl:
            em.sym[s] = em.emit(e)
        outputs = [(k, em.emit(e)) for k, e in enumerate(reduced)]
    except RecursionError as err:  # pragma: no cover
        raise Unsupported("expression too deep") from err
    # an output that is an input register or shared between slots is fine: OUT only reads
    ops, n_regs = _allocate(em.ops, outputs, max_regs)
    code = np.empty(len(ops), dtype=np.uint32)
    for k, (c, d, a, b) in enumerate(ops):
        if c == OP_CONST:
            a, b = b & 0xff, (b >> 8) & 0xff
        code[k] = (c & 0xff) | ((d & 0xff) << 8) | ((a & 0xff) << 16) | ((b & 0xff) << 24)
    return code, np.asarray(em.consts, dtype=np.float64), n_regs


def run_numpy(code, consts, n_regs, n_out, t, xx, yy):
    """Reference interpreter: the outputs of a program for arrays of points."""
    xx, yy = np.asarray(xx, dtype=float), np.asarray(yy, dtype=float)
    reg = [None] * n_regs
    reg[0], reg[1], reg[2] = np.full(xx.shape, float(t)), xx, yy
    out = [None] * n_out
    with np.errstate(all="ignore"):
        for w in np.asarray(code, dtype=np.uint32).tolist():
            op, d, a, b = w & 0xff, (w >> 8) & 0xff, (w >> 16) & 0xff, (w >> 24) & 0xff
            if op == OP_CONST:
                reg[d] = np.full(xx.shape, consts[a | (b << 8)])
                continue
            va = reg[a]
            if op == OP_OUT:
                out[d] = va
                continue
            if op == OP_ADD:
                r = va + reg[b]
            elif op == OP_SUB:
                r = va - reg[b]
            elif op == OP_MUL:
                r = va * reg[b]
            elif op == OP_DIV:
                r = va / reg[b]
            elif op == OP_NEG:
                r = -va
            elif op == OP_POWI:
                n = b - 256 if b > 127 else b
                r = np.ones_like(va)
                base, m = va, abs(n)
                while m:
                    if m & 1:
                        r = r * base
                    base = base * base
                    m >>= 1
                if n < 0:
                    r = 1.0 / r
            elif op == OP_POW:
                r = np.power(va, reg[b])
            elif op == OP_EXP:
                r = np.exp(va)
            elif op == OP_SIN:
                r = np.sin(va)
            elif op == OP_COS:
                r = np.cos(va)
            elif op == OP_SQRT:
                r = np.sqrt(va)
            elif op == OP_LOG:
                r = np.log(va)
            elif op == OP_ABS:
                r = np.abs(va)
            elif op == OP_SIGN:
                r = np.sign(va)
            elif op == OP_TANH:
                r = np.tanh(va)
            else:
                r = va
            reg[d] = r
    return out


NODE_OUTPUTS = ("cp", "dt_cp", "dx_cp", "dy_cp", "T", "dt_T", "dx_T", "dy_T", "lap_T", "cl", "dt_cl", "dx_cl", "dy_cl",
                "lap_cl", "cd", "dt_cd", "dx_cd", "dy_cd", "lap_cd", "cs", "dt_cs")
CELL_OUTPUTS = ("cp", "dt_cp", "cl", "T")


def lower_case(exprs_by_name, t, x, y):
    """The node and the cell program of a case from {name: expression} with the names of NODE_OUTPUTS."""
    node = lower([exprs_by_name[n] for n in NODE_OUTPUTS], t, x, y)
    cell = lower([exprs_by_name[n] for n in CELL_OUTPUTS], t, x, y)
    return node, cell
