"""NumPy stand-in for the third-party ``warp`` module (warp-lang 0.10.1) -- TEST INFRASTRUCTURE ONLY.

Why this exists: the reference's MPM substep is Warp DSL (``@wp.kernel`` Python functions in
/root/reference/warp_mpm/{mpm_utils,mpm_solver,mpm_data_structure}.py).  Warp itself is NVIDIA-only and
not installable here, but the kernel bodies are plain Python.  With this module on ``sys.path`` as
``warp`` the reference's three files are imported UNCHANGED from /root/reference and every kernel runs as
what Warp's CPU device makes of it: a serial ``for tid in range(dim)`` loop, fp32 arithmetic (NumPy
float32 scalars, one rounding per operation, no FMA contraction), C truncation for ``wp.int``.
``tests/golden/make_golden_ref.py`` uses it (build container only) to produce ``tests/golden/ref_*.npz``;
nothing else imports it and it never travels anywhere as part of the product.

What it is NOT: an emulation of Warp's code generator.  Only the API surface the reference's MPM path
touches is here.  Semantics that come from Warp's own sources (out of tree, restated from its docs /
headers from memory) and therefore stay UNPINNED:
  * ``svd3``  -- any A = U diag(s) V^T.  Two interchangeable conventions are provided (SVD_MODE):
                "lapack" (s >= 0 descending, U/V orthogonal of either handedness) and "rot"
                (U, V proper rotations, sign carried by the smallest singular value, as Warp's
                McAdams-style routine does).  The generator checks that the fixtures do not depend on
                the choice inside the tested domain (det F > 0).
  * ``qr3``   -- A = Q R with Q a proper rotation (Warp builds Q from a Givens quaternion) and R upper
                triangular; the signs of diag(R) are free and the reference fixes them itself
                (mpm_utils.py:112-123).  Two conventions (QR_MODE): "householder" and "gs".
  * ``mat * mat`` is the matrix product, ``mat33(v0, v1, v2)`` takes COLUMNS, ``mat33(9 scalars)`` is
    row-major, ``normalize(0) = 0``, ``mesh_eval_face_normal = normalize((q-p) x (r-p))``.
"""
from __future__ import annotations

import ctypes
import math
import time as _time
import sys
import types as _pytypes

import numpy as np

f32 = np.float32
_I32 = np.int32

SVD_MODE = "lapack"   # or "rot", "rot32", "mcadams" (the published fp32 algorithm behind wp.svd3)
QR_MODE = "householder"  # or "gs", "gs32", "givens" (Givens quaternions, as behind wp.qr3)

config = _pytypes.SimpleNamespace(mode="release", verify_cuda=False)


def init():
    return None


# ------------------------------------------------------------------------------------------ scalar types
class float32:  # noqa: N801  (dtype tag)
    _type_ = ctypes.c_float
    _np_ = np.float32
    _length_ = 1

    def __new__(cls, x=0.0):
        return np.float32(x)


class int32:  # noqa: N801
    _type_ = ctypes.c_int32
    _np_ = np.int32
    _length_ = 1

    def __new__(cls, x=0):
        return int(x)


class uint64:  # noqa: N801
    _type_ = ctypes.c_uint64
    _np_ = np.uint64
    _length_ = 1

    def __new__(cls, x=0):
        return int(x)


def float(x=0.0):  # noqa: A001  -- wp.float(i)
    return f32(x)


def int(x=0):  # noqa: A001  -- wp.int(x): C truncation towards zero
    return _b.int(x)  # Python's int() truncates towards zero, like the C cast Warp emits


import builtins as _b  # noqa: E402


def _scalar_kind(dtype):
    if dtype in (_b.float, float32, float):
        return float32
    if dtype in (_b.int, int32, int):
        return int32
    if dtype is uint64:
        return uint64
    return dtype


# ------------------------------------------------------------------------------------------ vec / mat
class _Vec:
    __array_ufunc__ = None  # let np.float32.__mul__(vec) fall through to our __rmul__
    __slots__ = ("a",)
    _type_ = ctypes.c_float
    _np_ = np.float32

    def __init__(self, *args):
        n = self._length_
        if len(args) == 0:
            self.a = np.zeros(n, f32)
        elif len(args) == 1 and isinstance(args[0], _Vec):
            self.a = args[0].a.copy()
        elif len(args) == 1 and isinstance(args[0], np.ndarray):
            self.a = args[0].astype(f32).reshape(n).copy()
        elif len(args) == 1 and isinstance(args[0], (list, tuple)):
            self.a = np.array(args[0], f32).reshape(n)
        elif len(args) == 1:
            self.a = np.full(n, args[0], f32)
        else:
            assert len(args) == n, (args, n)
            self.a = np.array(args, f32)

    @classmethod
    def _wrap(cls, a):
        o = cls.__new__(cls)
        o.a = a
        return o

    def __getitem__(self, i):
        return self.a[i]

    def __setitem__(self, i, v):
        self.a[i] = v

    def __len__(self):
        return self._length_

    def __iter__(self):
        return iter(self.a)

    def __add__(self, o):
        return self._wrap(self.a + o.a)

    def __sub__(self, o):
        return self._wrap(self.a - o.a)

    def __neg__(self):
        return self._wrap(-self.a)

    def __mul__(self, s):
        if isinstance(s, _Vec):
            return self._wrap(self.a * s.a)
        return self._wrap(self.a * f32(s))

    __rmul__ = __mul__

    def __truediv__(self, s):
        return self._wrap(self.a / f32(s))

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(repr(_b.float(x)) for x in self.a)})"

    def _assign(self, o):
        self.a[...] = o.a if isinstance(o, _Vec) else o


class vec2(_Vec):  # noqa: N801
    _length_ = 2
    _shape_ = (2,)


class vec3(_Vec):  # noqa: N801
    _length_ = 3
    _shape_ = (3,)


class vec4(_Vec):  # noqa: N801
    _length_ = 4
    _shape_ = (4,)


class quat(_Vec):  # noqa: N801
    _length_ = 4
    _shape_ = (4,)


class _Mat:
    __array_ufunc__ = None
    __slots__ = ("a",)
    _type_ = ctypes.c_float
    _np_ = np.float32

    def __init__(self, *args):
        n = self._shape_[0]
        if len(args) == 0:
            self.a = np.zeros((n, n), f32)
        elif len(args) == 1 and isinstance(args[0], _Mat):
            self.a = args[0].a.copy()
        elif len(args) == 1 and isinstance(args[0], np.ndarray):
            self.a = args[0].astype(f32).reshape(n, n).copy()
        elif len(args) == 1:
            self.a = np.full((n, n), args[0], f32)
        elif len(args) == n and isinstance(args[0], _Vec):
            # vector arguments are COLUMNS (deduced from w[axis, node] in mpm_utils.py:509-525)
            self.a = np.stack([v.a for v in args], axis=1).astype(f32)
        else:
            assert len(args) == n * n, (args,)
            self.a = np.array(args, f32).reshape(n, n)  # row-major scalars

    @classmethod
    def _wrap(cls, a):
        o = cls.__new__(cls)
        o.a = a
        return o

    def __getitem__(self, ij):
        return self.a[ij]

    def __setitem__(self, ij, v):
        self.a[ij] = v

    def __add__(self, o):
        return self._wrap(self.a + o.a)

    def __sub__(self, o):
        return self._wrap(self.a - o.a)

    def __neg__(self):
        return self._wrap(-self.a)

    def __mul__(self, o):
        if isinstance(o, _Mat):  # matrix product, accumulated k = 0, 1, 2 in fp32
            A, B = self.a, o.a
            acc = A[:, 0:1] * B[0:1, :]
            for k in range(1, A.shape[1]):
                acc = acc + A[:, k:k + 1] * B[k:k + 1, :]
            return self._wrap(acc)
        if isinstance(o, _Vec):  # matrix-vector: col0*b0 + col1*b1 + col2*b2
            A, b = self.a, o.a
            acc = A[:, 0] * b[0]
            for k in range(1, A.shape[1]):
                acc = acc + A[:, k] * b[k]
            return type(o)._wrap(acc)
        return self._wrap(self.a * f32(o))

    def __rmul__(self, s):
        return self._wrap(self.a * f32(s))

    def __truediv__(self, s):
        return self._wrap(self.a / f32(s))

    def __repr__(self):
        return f"{type(self).__name__}({self.a.tolist()})"

    def _assign(self, o):
        self.a[...] = o.a if isinstance(o, _Mat) else o


class mat22(_Mat):  # noqa: N801
    _shape_ = (2, 2)
    _length_ = 4


class mat33(_Mat):  # noqa: N801
    _shape_ = (3, 3)
    _length_ = 9


class mat44(_Mat):  # noqa: N801
    _shape_ = (4, 4)
    _length_ = 16


def _inner_shape(dtype):
    return tuple(getattr(dtype, "_shape_", ()))


# ------------------------------------------------------------------------------------------ arrays
class array:  # noqa: N801
    """wp.array: a NumPy array whose trailing axes are the vec / mat value type."""

    def __init__(self, data=None, dtype=None, shape=None, strides=None, length=0, ptr=None, capacity=0, device=None,
                 copy=True, owner=True, ndim=None, grad=None, requires_grad=False, **_):
        self.dtype = _scalar_kind(dtype) if dtype is not None else float32
        self.requires_grad = requires_grad
        self.grad = grad
        self.device = device
        self._data = None
        self.ndim = ndim if ndim is not None else 1
        if ptr is not None:  # alias foreign memory (from_torch_safe, warp_utils.py:75-86)
            if isinstance(shape, _b.int):
                shape = (shape,)
            inner = _inner_shape(self.dtype)
            full = tuple(shape) + inner
            n = _b.int(np.prod(full)) if len(full) else 1
            ct = self.dtype._type_
            buf = (ct * max(n, 1)).from_address(ptr) if n > 0 else None
            if n > 0:
                self._data = np.ctypeslib.as_array(buf).reshape(full)
            else:
                self._data = np.zeros(full, self.dtype._np_)
            self.ndim = len(tuple(shape))
        elif data is not None:
            self._data = np.array(data, dtype=self.dtype._np_, copy=True)
            self.ndim = self._data.ndim - len(_inner_shape(self.dtype))

    @property
    def shape(self):
        d = self._data
        k = len(_inner_shape(self.dtype))
        return tuple(d.shape[: d.ndim - k])

    @property
    def size(self):
        return _b.int(np.prod(self.shape))

    @property
    def ptr(self):
        return self._data.ctypes.data

    def numpy(self):
        return self._data

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, idx):
        v = self._data[idx]
        dt = self.dtype
        if dt is float32 or dt is int32 or dt is uint64:
            return v
        return dt._wrap(v.copy())

    def __setitem__(self, idx, val):
        if isinstance(val, (_Vec, _Mat)):
            self._data[idx] = val.a
        else:
            self._data[idx] = val

    def zero_(self):
        self._data[...] = 0

    def fill_(self, v):
        self._data[...] = v.a if isinstance(v, (_Vec, _Mat)) else v


def _alloc(shape, dtype):
    dtype = _scalar_kind(dtype)
    if isinstance(shape, (_b.int, np.integer)):
        shape = (_b.int(shape),)
    a = array(dtype=dtype)
    a._data = np.zeros(tuple(shape) + _inner_shape(dtype), dtype._np_)
    a.ndim = len(tuple(shape))
    return a


def zeros(shape=None, dtype=float32, device=None, requires_grad=False, **_):
    a = _alloc(shape, dtype)
    a.requires_grad = requires_grad
    a.device = device
    return a


empty = zeros


def zeros_like(src, requires_grad=False, **_):
    a = array(dtype=src.dtype)
    a._data = np.zeros_like(src._data)
    a.ndim = src.ndim
    a.requires_grad = requires_grad
    return a


def clone(src, requires_grad=False, **_):
    a = zeros_like(src, requires_grad)
    a._data[...] = src._data
    return a


def copy(dest, src, **_):
    dest._data[...] = src._data.reshape(dest._data.shape)


def from_numpy(arr, dtype=None, device=None, requires_grad=False, **_):
    arr = np.asarray(arr)
    if dtype is None:
        dtype = int32 if np.issubdtype(arr.dtype, np.integer) else float32
    dtype = _scalar_kind(dtype)
    inner = _inner_shape(dtype)
    a = array(dtype=dtype)
    data = np.array(arr, dtype=dtype._np_, copy=True)
    if inner and tuple(data.shape[-len(inner):]) != inner:
        data = data.reshape((-1,) + inner)
    a._data = data
    a.ndim = data.ndim - len(inner)
    a.requires_grad = requires_grad
    a.device = device
    return a


# ------------------------------------------------------------------------------------------ decorators / launch
_TID = None


def tid():
    return _TID


class Kernel:
    def __init__(self, fn):
        self.func = fn
        self.key = fn.__name__
        ann = dict(getattr(fn, "__annotations__", {}))
        ann.pop("return", None)
        self.arg_names = list(fn.__code__.co_varnames[: fn.__code__.co_argcount])
        self.arg_types = [ann.get(n) for n in self.arg_names]

    def __call__(self, *a, **k):
        raise RuntimeError("kernels are launched with wp.launch")


def kernel(fn=None, **_):
    if fn is None:  # "@wp.kernel()" (mpm_utils.py:1515)
        return lambda f: Kernel(f)
    return Kernel(fn)


def func(fn):
    return fn


def _default_for(tp):
    if tp in (_b.float, float32, float):
        return f32(0.0)
    if tp in (_b.int, int32, int, uint64):
        return 0
    if isinstance(tp, type) and issubclass(tp, (_Vec, _Mat)):
        return tp()
    return None


def _cast_for(tp, v):
    if v is None:
        return v
    if tp in (_b.float, float32, float):
        return f32(v)
    if tp in (_b.int, int32, int, uint64):
        return _b.int(v)
    if isinstance(tp, type) and issubclass(tp, (_Vec, _Mat)):
        return v if isinstance(v, tp) else tp(*v) if isinstance(v, (list, tuple)) and issubclass(tp, _Vec) else tp(v)
    return v


def struct(cls):
    """@wp.struct: typed fields (float -> fp32, int -> int, vec/mat values, arrays by reference)."""
    ann = dict(getattr(cls, "__annotations__", {}))
    orig_init = cls.__init__ if "__init__" in cls.__dict__ else None

    def __init__(self, *a, **k):
        for name, tp in ann.items():
            object.__setattr__(self, name, _default_for(tp))
        if orig_init is not None:
            orig_init(self, *a, **k)

    def __setattr__(self, name, value):
        tp = ann.get(name)
        object.__setattr__(self, name, _cast_for(tp, value) if tp is not None else value)

    cls.__init__ = __init__
    cls.__setattr__ = __setattr__
    cls._wp_struct_ = True
    return cls


def launch(kernel, dim, inputs=(), outputs=(), device=None, **_):  # noqa: A002
    """Warp's CPU device: one thread, tids in C order (x outer ... z inner for 3-D launches)."""
    global _TID
    args = list(inputs) + list(outputs)
    k = kernel
    assert isinstance(k, Kernel), k
    assert len(args) == len(k.arg_types), (k.key, len(args), len(k.arg_types))
    cast = []
    for tp, v in zip(k.arg_types, args):
        if tp in (_b.float, float32, float):
            v = f32(v)
        elif tp in (_b.int, int32, int):
            v = _b.int(v)
        cast.append(v)
    fn = k.func
    if isinstance(dim, (_b.int, np.integer)):
        for t in range(_b.int(dim)):
            _TID = t
            fn(*cast)
    else:
        dims = tuple(_b.int(d) for d in dim)
        if len(dims) == 1:
            for t in range(dims[0]):
                _TID = t
                fn(*cast)
        elif len(dims) == 2:
            for i in range(dims[0]):
                for j in range(dims[1]):
                    _TID = (i, j)
                    fn(*cast)
        else:
            for i in range(dims[0]):
                for j in range(dims[1]):
                    for l in range(dims[2]):
                        _TID = (i, j, l)
                        fn(*cast)
    _TID = None


class ScopedTimer:
    def __init__(self, name, synchronize=False, print=True, dict=None, **_):  # noqa: A002
        self.name, self.dict = name, dict

    def __enter__(self):
        self.t0 = _time.perf_counter()
        return self

    def __exit__(self, *exc):
        if self.dict is not None:
            self.dict.setdefault(self.name, []).append((_time.perf_counter() - self.t0) * 1000.0)
        return False


def synchronize():
    return None


def get_device(name=None):
    return name or "cpu"


# ------------------------------------------------------------------------------------------ builtins
def _s(x):
    return x if isinstance(x, np.floating) else f32(x)


def sqrt(x):
    return np.sqrt(_s(x))


def log(x):
    with np.errstate(all="ignore"):
        return np.log(_s(x))


def exp(x):
    with np.errstate(all="ignore"):
        return np.exp(_s(x))


def sin(x):
    return np.sin(_s(x))


def cos(x):
    return np.cos(_s(x))


def tan(x):
    return np.tan(_s(x))


def acos(x):
    return np.arccos(np.clip(_s(x), f32(-1.0), f32(1.0)))  # Warp clamps the argument


def pow(x, y):  # noqa: A001
    return np.power(_s(x), _s(y))


def abs(x):  # noqa: A001
    return np.abs(x) if not isinstance(x, (_b.int, _b.float)) else _b.abs(x)


def min(a, b):  # noqa: A001
    return a if a < b else b


def max(a, b):  # noqa: A001
    return a if a > b else b


def clamp(x, lo, hi):
    return min(max(x, lo), hi)


def dot(a, b):
    p = a.a * b.a
    acc = p[0]
    for k in range(1, p.shape[0]):
        acc = acc + p[k]
    return acc


def length(a):
    return np.sqrt(dot(a, a))


def length_sq(a):
    return dot(a, a)


def normalize(a):
    l = length(a)
    if l > f32(0.0):
        return type(a)._wrap(a.a / l)
    return type(a)()


def cross(a, b):
    x, y = a.a, b.a
    return vec3._wrap(np.array([x[1] * y[2] - x[2] * y[1], x[2] * y[0] - x[0] * y[2], x[0] * y[1] - x[1] * y[0]], f32))


def cw_mul(a, b):
    return type(a)._wrap(a.a * b.a)


def cw_div(a, b):
    return type(a)._wrap(a.a / b.a)


def outer(a, b):
    return mat33._wrap(a.a[:, None] * b.a[None, :]) if a.a.shape[0] == 3 else mat22._wrap(a.a[:, None] * b.a[None, :])


def transpose(m):
    return type(m)._wrap(m.a.T.copy())


def diag(v):
    n = v.a.shape[0]
    out = np.zeros((n, n), f32)
    out[np.arange(n), np.arange(n)] = v.a
    return (mat33 if n == 3 else mat22)._wrap(out)


def trace(m):
    a = m.a
    acc = a[0, 0]
    for k in range(1, a.shape[0]):
        acc = acc + a[k, k]
    return acc


def ddot(a, b):
    return (a.a * b.a).sum(dtype=f32)


def determinant(m):
    a = m.a
    if a.shape[0] == 2:
        return a[0, 0] * a[1, 1] - a[0, 1] * a[1, 0]
    # dot(col0, cross(col1, col2))
    c0, c1, c2 = vec3._wrap(a[:, 0].copy()), vec3._wrap(a[:, 1].copy()), vec3._wrap(a[:, 2].copy())
    return dot(c0, cross(c1, c2))


def inverse(m):
    return type(m)._wrap(np.linalg.inv(m.a.astype(np.float64)).astype(f32))


def add(a, b):
    return a + b


def sub(a, b):
    return a - b


def mul(a, b):
    return a * b


def identity(n=3, dtype=float32):
    return (mat33 if n == 3 else mat22)._wrap(np.eye(n, dtype=f32))


def _idx(args):
    return args[0] if len(args) == 1 else tuple(args)


def atomic_add(arr, *args):
    *idx, val = args
    i = _idx(idx)
    old = arr[i]
    if isinstance(val, (_Vec, _Mat)):
        arr._data[i] = arr._data[i] + val.a
    else:
        arr._data[i] = arr._data[i] + arr.dtype._np_(val)
    return old


def atomic_sub(arr, *args):
    *idx, val = args
    i = _idx(idx)
    old = arr[i]
    if isinstance(val, (_Vec, _Mat)):
        arr._data[i] = arr._data[i] - val.a
    else:
        arr._data[i] = arr._data[i] - arr.dtype._np_(val)
    return old


# -- svd3 / qr3: out-parameters are written in place ------------------------------------------------------
def _svd_lapack(A):
    U, s, Vt = np.linalg.svd(A.astype(np.float64))
    return U, s, Vt.T


def _svd_rot(A):
    U, s, V = _svd_lapack(A)
    if np.linalg.det(U) < 0:
        U[:, 2] = -U[:, 2]
        s[2] = -s[2]
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
        s[2] = -s[2]
    return U, s, V


def _svd_rot32(A):
    """Same convention as "rot", but computed in fp32 ARITHMETIC (one-sided Jacobi on the columns; numpy.linalg would
    silently compute in double): accurate to fp32 rounding only, like any fp32 implementation (Warp's included).  Used
    to measure how sensitive the reference's trajectories are to that."""
    B = A.astype(f32).copy()
    V = np.eye(3, dtype=f32)
    for _ in range(8):
        for p, q in ((0, 1), (0, 2), (1, 2)):
            a = (B[:, p] * B[:, p]).sum(dtype=f32)
            b = (B[:, q] * B[:, q]).sum(dtype=f32)
            c = (B[:, p] * B[:, q]).sum(dtype=f32)
            if np.abs(c) <= f32(1e-12) * np.sqrt(a * b):
                continue
            zeta = (b - a) / (f32(2.0) * c)
            t = np.sign(zeta) / (np.abs(zeta) + np.sqrt(f32(1.0) + zeta * zeta)) if zeta != 0 else f32(1.0)
            cs = f32(1.0) / np.sqrt(f32(1.0) + t * t)
            sn = cs * t
            bp, bq = B[:, p].copy(), B[:, q].copy()
            B[:, p], B[:, q] = cs * bp - sn * bq, sn * bp + cs * bq
            vp, vq = V[:, p].copy(), V[:, q].copy()
            V[:, p], V[:, q] = cs * vp - sn * vq, sn * vp + cs * vq
    s = np.sqrt((B * B).sum(axis=0, dtype=f32)).astype(f32)
    order = np.argsort(-s)
    B, V, s = B[:, order], V[:, order].copy(), s[order].copy()
    U = np.zeros((3, 3), f32)
    for k in range(3):
        U[:, k] = B[:, k] / s[k] if s[k] > f32(1e-30) else 0
    if s[2] <= f32(1e-30):  # rank-deficient (the cloth model's zero-padded 2x2): complete the basis
        U[:, 2] = np.cross(U[:, 0], U[:, 1])
    if np.linalg.det(U.astype(np.float64)) < 0:
        U[:, 2] = -U[:, 2]
        s[2] = -s[2]
    if np.linalg.det(V.astype(np.float64)) < 0:
        V[:, 2] = -V[:, 2]
        s[2] = -s[2]
    return U, s, V


# -- the PUBLISHED algorithm behind wp.svd3 / wp.qr3 (round 4) --------------------------------------------------------------
# A. McAdams, A. Selle, R. Tamstorf, J. Teran, E. Sifakis, "Computing the Singular Value Decomposition of 3x3 matrices with
# minimal branching and elementary floating point operations", UW-Madison TR1690 (2011): Jacobi eigenanalysis of A^T A with
# approximate Givens quaternions (4 sweeps of 3 conjugations), singular values sorted by column norm with sign-carrying swaps,
# then a QR of A V by three Givens quaternions (U = Q, sigma = diag(R)).  Warp's native svd3 / qr3 follow this scheme (its
# sources are out of tree: this is the paper's algorithm restated in fp32 NumPy scalars, one rounding per operation -- not
# Warp's code, and Q is accumulated by applying the three rotations instead of the closed form).
_MC_GAMMA, _MC_CSTAR, _MC_SSTAR, _MC_EPS = f32(5.828427124), f32(0.923879532), f32(0.3826834323), f32(1e-6)


def _mc_rsqrt(x):
    return f32(1.0) / np.sqrt(f32(x))


def _mc_approx_givens(a11, a12, a22):
    ch, sh = f32(2.0) * (a11 - a22), a12
    b = _MC_GAMMA * sh * sh < ch * ch
    w = _mc_rsqrt(ch * ch + sh * sh) if (ch * ch + sh * sh) > 0 else f32(0.0)
    return (w * ch, w * sh) if b else (_MC_CSTAR, _MC_SSTAR)


def _mc_jacobi(S):
    """S symmetric 3x3 (fp32) -> unit quaternion (x, y, z, w) of V with V^T S V ~ diagonal."""
    s11, s21, s22, s31, s32, s33 = S[0, 0], S[1, 0], S[1, 1], S[2, 0], S[2, 1], S[2, 2]
    q = [f32(0.0), f32(0.0), f32(0.0), f32(1.0)]
    for _ in range(4):
        for (x, y, z) in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
            ch, sh = _mc_approx_givens(s11, s21, s22)
            scale = ch * ch + sh * sh
            a, b = (ch * ch - sh * sh) / scale, (f32(2.0) * sh * ch) / scale
            t11 = a * (a * s11 + b * s21) + b * (a * s21 + b * s22)
            t21 = a * (-b * s11 + a * s21) + b * (-b * s21 + a * s22)
            t22 = -b * (-b * s11 + a * s21) + a * (-b * s21 + a * s22)
            t31 = a * s31 + b * s32
            t32 = -b * s31 + a * s32
            t33 = s33
            tmp = [q[0] * sh, q[1] * sh, q[2] * sh]
            sh = sh * q[3]
            q = [q[0] * ch, q[1] * ch, q[2] * ch, q[3] * ch]
            q[z] = q[z] + sh
            q[3] = q[3] - tmp[z]
            q[x] = q[x] + tmp[y]
            q[y] = q[y] - tmp[x]
            s11, s21, s22, s31, s32, s33 = t22, t32, t33, t21, t31, t11   # re-arranged for the next pair
    n = _mc_rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    return [c * n for c in q]


def _mc_quat_to_mat(q):
    x, y, z, w = q
    two = f32(2.0)
    return np.array([[f32(1) - two * (y * y + z * z), two * (x * y - w * z), two * (x * z + w * y)],
                     [two * (x * y + w * z), f32(1) - two * (x * x + z * z), two * (y * z - w * x)],
                     [two * (x * z - w * y), two * (y * z + w * x), f32(1) - two * (x * x + y * y)]], f32)


def _mc_qr_givens_quat(a1, a2):
    rho = np.sqrt(a1 * a1 + a2 * a2)
    sh = a2 if rho > _MC_EPS else f32(0.0)
    ch = np.abs(a1) + max(rho, _MC_EPS)
    if a1 < 0:
        sh, ch = ch, sh
    w = _mc_rsqrt(ch * ch + sh * sh)
    return ch * w, sh * w


def _mc_rot(i, j, ch, sh):
    """Rotation matrix of the Givens quaternion (ch, sh about the axis perpendicular to the (i, j) plane)."""
    a, b = f32(1.0) - f32(2.0) * sh * sh, f32(2.0) * ch * sh
    G = np.eye(3, dtype=f32)
    G[i, i], G[i, j], G[j, i], G[j, j] = a, -b, b, a
    return G


def _mm32(A, B):
    C = np.zeros((3, 3), f32)
    for i in range(3):
        for j in range(3):
            C[i, j] = A[i, 0] * B[0, j] + A[i, 1] * B[1, j] + A[i, 2] * B[2, j]
    return C


def _qr_givens(A):
    """A = Q R, Q a proper rotation built from three Givens quaternions zeroing (2,1), (3,1), (3,2) in that order."""
    R = A.astype(f32).copy()
    Q = np.eye(3, dtype=f32)
    for (i, j) in ((0, 1), (0, 2), (1, 2)):
        ch, sh = _mc_qr_givens_quat(R[i, i], R[j, i])
        G = _mc_rot(i, j, ch, sh)
        R = _mm32(G.T.copy(), R)
        Q = _mm32(Q, G)
    return Q, R


def _svd_mcadams(A):
    A = A.astype(f32)
    V = _mc_quat_to_mat(_mc_jacobi(_mm32(A.T.copy(), A)))
    B = _mm32(A, V)
    rho = [B[0, k] * B[0, k] + B[1, k] * B[1, k] + B[2, k] * B[2, k] for k in range(3)]

    def neg_swap(c, k, l):
        if c:
            for M in (B, V):
                t = -M[:, k].copy()
                M[:, k] = M[:, l]
                M[:, l] = t
            rho[k], rho[l] = rho[l], rho[k]
    neg_swap(rho[0] < rho[1], 0, 1)
    neg_swap(rho[0] < rho[2], 0, 2)
    neg_swap(rho[1] < rho[2], 1, 2)
    U, R = _qr_givens(B)
    return U, np.array([R[0, 0], R[1, 1], R[2, 2]], f32), V



def svd3(A, U, sigma, V):
    u, s, v = {"rot": _svd_rot, "rot32": _svd_rot32, "mcadams": _svd_mcadams}.get(SVD_MODE, _svd_lapack)(A.a)
    U._assign(u.astype(f32))
    sigma._assign(s.astype(f32))
    V._assign(v.astype(f32))


def _qr_householder(A):
    Q, R = np.linalg.qr(A.astype(np.float64))
    if np.linalg.det(Q) < 0:  # Warp's Q comes from a quaternion: always a proper rotation
        Q[:, 2] = -Q[:, 2]
        R[2, :] = -R[2, :]
    return Q, R


def _qr_gs(A):
    A = A.astype(np.float64)
    q1 = A[:, 0] / np.linalg.norm(A[:, 0])
    u2 = A[:, 1] - (q1 @ A[:, 1]) * q1
    q2 = u2 / np.linalg.norm(u2)
    q3 = np.cross(q1, q2)
    Q = np.stack([q1, q2, q3], 1)
    R = np.triu(Q.T @ A)
    return Q, R


def _qr_gs32(A):
    """Gram-Schmidt in fp32 arithmetic (fp32-rounding-level accuracy, see _svd_rot32)."""
    A = A.astype(f32)
    q1 = A[:, 0] / np.sqrt((A[:, 0] * A[:, 0]).sum(dtype=f32))
    u2 = A[:, 1] - (q1 * A[:, 1]).sum(dtype=f32) * q1
    q2 = u2 / np.sqrt((u2 * u2).sum(dtype=f32))
    q3 = np.cross(q1, q2).astype(f32)
    Q = np.stack([q1, q2, q3], 1)
    R = np.triu((Q.T @ A).astype(f32))
    return Q, R


def qr3(A, Q, R):
    q, r = {"gs": _qr_gs, "gs32": _qr_gs32, "givens": _qr_givens}.get(QR_MODE, _qr_householder)(A.a)
    Q._assign(q.astype(f32))
    R._assign(np.triu(r).astype(f32))


# -- meshes --------------------------------------------------------------------------------------------
_MESHES = {}


class Mesh:
    def __init__(self, points=None, velocities=None, indices=None, **_):
        self.points, self.velocities, self.indices = points, velocities, indices
        self.id = 0x1000 + len(_MESHES)
        _MESHES[self.id] = self

    def refit(self):
        return None


def mesh_get(mesh_id):
    return _MESHES[_b.int(mesh_id)]


def mesh_eval_face_normal(mesh_id, face):
    m = _MESHES[_b.int(mesh_id)]
    i, j, k = (_b.int(m.indices[3 * face + c]) for c in range(3))
    p, q, r = m.points[i], m.points[j], m.points[k]
    return normalize(cross(q - p, r - p))


# ------------------------------------------------------------------------------------------ submodules
from . import types, context  # noqa: E402,F401
from . import torch as _wp_torch  # noqa: E402
from .torch import from_torch, to_torch  # noqa: E402,F401

torch = _wp_torch
