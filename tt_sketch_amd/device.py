"""Device arrays and the strided contraction front end of libttsk.

``DevArray`` is a strided view on HBM owned by the library (ttsk_malloc).  It
supports the few NumPy idioms the sketch path uses -- ``.T``, ``transpose``,
``reshape``, basic slicing, ``np.asarray(x)`` -- as zero-copy views, so that
``Tensor.T`` (tensor.py:311-313 and friends in the reference), the
``rank_min:rank_max`` column slices (tensor_train_drm.py:88) and the
unfoldings (utils.py:63-83) never move data.

``contract("ij,jkl->ikl", A, B)`` lowers a two-operand einsum onto the MFMA
contraction kernel (ttsk_gemm): index letters are classified into batch / M /
N / K groups and merged into single strided dimensions.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _native as nat

_F64 = np.dtype(np.float64)


# Device buffers are recycled instead of hipFree'd: hipMalloc / hipFree of GB-sized buffers cost
# 10-100 ms each (the dense-tensor path allocates several per sketch and was swinging between 80 and
# 500 ms per call), and every hipFree is a device-wide wait (28 of the 35 ms of an orthogonal_sketch
# at the north-star shape went there, 60 small temporaries).
#
# Reuse is stream-ordered.  A released buffer is tagged with the library streams that had work in
# flight at that moment (``nat.dirty_snapshot()``: a superset of the streams that can still be reading
# or writing it).  It is handed out again only when every one of those streams has been drained since,
# or is the stream the new owner says it will use first (``DevArray.empty(..., stream=)``, default 0):
# work queued later on the same stream is ordered behind the old owner's.  So the single-stream paths
# recycle at once, and whatever is released inside a multi-stream region (assemble_sketched_tt,
# pinv_dev_many, contract(..., stream=k) temporaries) waits for that region's closing sync().
_POOL_MIN = 8 << 20          # bytes; from here on sizes are rounded to 2 MB instead of powers of two
_POOL_CAP = 48 << 30         # bytes kept at most (large classes)
_SMALL_CAP = 2 << 30         # bytes kept at most (small classes)
_SCAN = 16                   # entries of a size class examined per request
_pool: "dict[int, list]" = {}   # size class -> [(tag, ptr), ...], most recently released last
_pool_bytes = {True: 0, False: 0}     # large / small


def _prod(shape) -> int:
    n = 1
    for x in shape:
        n *= int(x)
    return n


def _size_class(nbytes: int) -> int:
    if nbytes >= _POOL_MIN:
        return (int(nbytes) + (2 << 20) - 1) & ~((2 << 20) - 1)
    n = 256
    while n < nbytes:
        n <<= 1
    return n


def _reusable(tag: dict, stream: int) -> bool:
    for s, gen in tag.items():
        if s != stream and not nat.drained_since(s, gen):
            return False
    return True


def _pool_take(size: int, stream: int):
    lst = _pool.get(size)
    if lst:
        for i in range(len(lst) - 1, max(len(lst) - 1 - _SCAN, -1), -1):
            if _reusable(lst[i][0], stream):
                _, ptr = lst.pop(i)
                _pool_bytes[size >= _POOL_MIN] -= size
                return ptr
    return None


def _pool_give(size: int, ptr: int) -> None:
    big = size >= _POOL_MIN
    cap = _POOL_CAP if big else _SMALL_CAP
    if not big and _pool_bytes[False] + size > cap:
        nat.lib().ttsk_free(ctypes.c_void_p(ptr))          # hipFree waits for the device itself
        return
    _pool.setdefault(size, []).append((nat.dirty_snapshot(), ptr))
    _pool_bytes[big] += size
    while big and _pool_bytes[True] > cap:          # drop the largest class first
        top = max((k for k, v in _pool.items() if v and k >= _POOL_MIN), default=None)
        if top is None:
            break
        _, old = _pool[top].pop(0)
        _pool_bytes[True] -= top
        nat.lib().ttsk_free(ctypes.c_void_p(old))


def release_cached(min_bytes: int = _POOL_MIN) -> int:
    """hipFree every pooled buffer of at least ``min_bytes`` (after a device-wide sync); returns the bytes
    released.  For callers that move between working sets of very different sizes (bench.py's sub-records)."""
    import gc
    gc.collect()
    nat.call("ttsk_sync", -1)
    freed = 0
    for size in [k for k in _pool if k >= min_bytes]:
        for _, ptr in _pool.pop(size):
            nat.lib().ttsk_free(ctypes.c_void_p(ptr))
            _pool_bytes[size >= _POOL_MIN] -= size
            freed += size
    return freed


class _Buffer:
    """Owns one ttsk_malloc allocation (recycled through the pool above)."""
    __slots__ = ("ptr", "nbytes", "_pooled")

    def __init__(self, nbytes: int, stream: int = 0):
        self.nbytes = int(nbytes)
        self._pooled = _size_class(self.nbytes)
        got = _pool_take(self._pooled, int(stream))
        if got is not None:
            self.ptr = got
            return
        p = ctypes.c_void_p()
        nat.call("ttsk_malloc", ctypes.byref(p), ctypes.c_size_t(self._pooled))
        self.ptr = p.value

    def __del__(self):
        try:
            if self.ptr:
                _pool_give(self._pooled, self.ptr)
        except Exception:  # interpreter shutdown
            pass
        self.ptr = None


def _c_strides(shape):
    st, acc = [], 1
    for n in reversed(shape):
        st.append(acc)
        acc *= int(n)
    return tuple(reversed(st))


class DevArray:
    """Strided fp64 / int64 array in device memory (strides in elements)."""
    __slots__ = ("buf", "offset", "shape", "strides", "dtype")
    __array_priority__ = 100

    def __init__(self, buf, offset, shape, strides, dtype=_F64):
        self.buf = buf
        self.offset = int(offset)
        self.shape = tuple(int(s) for s in shape)
        self.strides = tuple(int(s) for s in strides)
        self.dtype = np.dtype(dtype)

    # ---- construction
    @classmethod
    def empty(cls, shape, dtype=_F64, stream=0) -> "DevArray":
        """``stream``: the library stream that touches the array first (see the pool note above)."""
        shape = tuple(int(s) for s in (shape if np.ndim(shape) else (shape,)))
        n = _prod(shape)
        return cls(_Buffer(n * np.dtype(dtype).itemsize, stream), 0, shape, _c_strides(shape), dtype)

    @classmethod
    def zeros(cls, shape, dtype=_F64, stream=0) -> "DevArray":
        a = cls.empty(shape, dtype, stream)
        if a.size:
            nat.call("ttsk_memset", ctypes.c_void_p(a.ptr), 0, ctypes.c_size_t(a.size * a.dtype.itemsize), stream)
        return a

    @classmethod
    def from_host(cls, arr, dtype=None, stream=0) -> "DevArray":
        arr = np.ascontiguousarray(arr, dtype=dtype if dtype is not None else getattr(arr, "dtype", None))
        if arr.dtype not in (np.float64, np.int64, np.uint64):
            arr = arr.astype(np.float64)
        a = cls.empty(arr.shape, arr.dtype, stream)
        if arr.size:
            nat.call("ttsk_h2d", ctypes.c_void_p(a.ptr), ctypes.c_void_p(arr.ctypes.data),
                     ctypes.c_size_t(arr.nbytes), stream)
        return a

    # ---- basic properties
    @property
    def ptr(self) -> int:
        return self.buf.ptr + self.offset * self.dtype.itemsize

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return _prod(self.shape)

    def is_contiguous(self) -> bool:
        exp = 1
        for n, s in zip(reversed(self.shape), reversed(self.strides)):
            if n != 1 and s != exp:
                return False
            exp *= n
        return True

    def __repr__(self):
        return f"<DevArray {self.dtype} shape={self.shape} strides={self.strides}>"

    def __len__(self):
        return self.shape[0]

    # ---- views
    @property
    def T(self) -> "DevArray":
        return DevArray(self.buf, self.offset, self.shape[::-1], self.strides[::-1], self.dtype)

    def transpose(self, *axes) -> "DevArray":
        if len(axes) == 1 and not np.isscalar(axes[0]):
            axes = tuple(axes[0])
        if not axes:
            return self.T
        return DevArray(self.buf, self.offset, [self.shape[a] for a in axes],
                        [self.strides[a] for a in axes], self.dtype)

    def reshape(self, *shape) -> "DevArray":
        if len(shape) == 1 and not np.isscalar(shape[0]):
            shape = tuple(shape[0])
        shape = list(int(s) for s in shape)
        if -1 in shape:
            k = shape.index(-1)
            rest = _prod([s for i, s in enumerate(shape) if i != k])
            shape[k] = self.size // rest if rest else 0
        if _prod(shape) != self.size:
            raise ValueError(f"cannot reshape {self.shape} into {tuple(shape)}")
        st = _view_strides(self.shape, self.strides, shape)
        if st is None:
            return self.contiguous().reshape(shape)
        return DevArray(self.buf, self.offset, shape, st, self.dtype)

    def __getitem__(self, key) -> "DevArray":
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            i = [k is Ellipsis for k in key].index(True)
            fill = (slice(None),) * (self.ndim - (len(key) - 1 - sum(k is None for k in key)))
            key = key[:i] + fill + key[i + 1:]
        off, shape, strides, ax = self.offset, [], [], 0
        for k in key:
            if k is None:
                shape.append(1)
                strides.append(0)
                continue
            n, s = self.shape[ax], self.strides[ax]
            if isinstance(k, slice):
                a, b, step = k.indices(n)
                if step != 1:
                    raise IndexError("DevArray supports unit-step slices only")
                b = max(a, b)
                off += a * s
                shape.append(b - a)
                strides.append(s)
            else:
                k = int(k)
                if k < 0:
                    k += n
                if not 0 <= k < n:
                    raise IndexError("index out of range")
                off += k * s
            ax += 1
        shape += self.shape[ax:]
        strides += self.strides[ax:]
        return DevArray(self.buf, off, shape, strides, self.dtype)

    # ---- data movement
    def contiguous(self, stream=0) -> "DevArray":
        if self.is_contiguous():
            return self
        if self.dtype != _F64:
            raise ValueError("strided copies are implemented for fp64 only")
        out = DevArray.empty(self.shape, self.dtype, stream)
        copy_into(out, self, stream)
        return out

    def copy(self, stream=0) -> "DevArray":
        out = DevArray.empty(self.shape, self.dtype, stream)
        if self.size:
            if self.is_contiguous():
                nat.call("ttsk_d2d", ctypes.c_void_p(out.ptr), ctypes.c_void_p(self.ptr),
                         ctypes.c_size_t(self.size * self.dtype.itemsize), stream)
            else:
                copy_into(out, self, stream)
        return out

    def get(self, stream=0) -> np.ndarray:
        src = self.contiguous(stream)
        out = np.empty(self.shape, dtype=self.dtype)
        if out.size:
            nat.call("ttsk_d2h", ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(src.ptr),
                     ctypes.c_size_t(out.nbytes), stream)
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a if dtype is None else a.astype(dtype, copy=False)

    def fill_zero(self, stream=0):
        if not self.is_contiguous():
            raise ValueError("fill_zero needs a contiguous array")
        if self.size:
            nat.call("ttsk_memset", ctypes.c_void_p(self.ptr), 0,
                     ctypes.c_size_t(self.size * self.dtype.itemsize), stream)
        return self


def _view_strides(shape, strides, new_shape) -> Optional[Tuple[int, ...]]:
    """Strides of a reshape that needs no copy, or None (NumPy's no-copy rule)."""
    old = [(n, s) for n, s in zip(shape, strides) if n != 1]
    new_strides = [0] * len(new_shape)
    if _prod(new_shape) == 0:
        return _c_strides(new_shape)
    oi = 0
    ni = 0
    nn = len(new_shape)
    while ni < nn:
        if new_shape[ni] == 1:
            new_strides[ni] = 0
            ni += 1
            continue
        if oi >= len(old):
            return None
        # gather a run of old dims and new dims with equal products
        op, np_ = old[oi][0], new_shape[ni]
        oj, nj = oi + 1, ni + 1
        while op != np_:
            if op < np_:
                if oj >= len(old):
                    return None
                op *= old[oj][0]
                oj += 1
            else:
                while nj < nn and new_shape[nj] == 1:
                    nj += 1
                if nj >= nn:
                    return None
                np_ *= new_shape[nj]
                nj += 1
        # old dims oi..oj must be mutually contiguous
        for k in range(oi, oj - 1):
            if old[k][1] != old[k + 1][1] * old[k + 1][0]:
                return None
        st = old[oj - 1][1]
        for k in range(nj - 1, ni - 1, -1):
            new_strides[k] = st if new_shape[k] != 1 else 0
            st *= new_shape[k]
        oi, ni = oj, nj
    if oi != len(old):
        return None
    return tuple(new_strides)


def copy_into(dst: DevArray, src: DevArray, stream=0):
    if dst.shape != src.shape:
        raise ValueError(f"copy_into: shape {src.shape} -> {dst.shape}")
    if src.size == 0:
        return
    # drop unit dims, keep at most 5
    dims = [(n, d, s) for n, d, s in zip(src.shape, dst.strides, src.strides) if n != 1]
    merged = []
    for n, d, s in dims:  # merge neighbours that are contiguous in both
        if merged and merged[-1][1] == d * n and merged[-1][2] == s * n:
            pn, pd, ps = merged[-1]
            merged[-1] = (pn * n, d, s)
        else:
            merged.append((n, d, s))
    if len(merged) > 5:
        raise ValueError("copy_into supports at most 5 non-mergeable dimensions")
    nd = len(merged)
    A = (ctypes.c_int64 * max(nd, 1))
    nat.call("ttsk_copy_strided", ctypes.c_void_p(dst.ptr), ctypes.c_void_p(src.ptr), nd,
             A(*[m[0] for m in merged]), A(*[m[1] for m in merged]), A(*[m[2] for m in merged]), stream)


def as_dev(x, stream=0) -> DevArray:
    """DevArray as is; NumPy arrays (e.g. from a user-defined DRM) are uploaded."""
    if isinstance(x, DevArray):
        return x
    return DevArray.from_host(np.asarray(x, dtype=np.float64), stream=stream)


def to_host(x) -> np.ndarray:
    return x.get() if isinstance(x, DevArray) else np.asarray(x)


def sync(stream=-1):
    nat.call("ttsk_sync", stream)


# ------------------------------------------------------------------ contraction
def _merge(letters, size, *stride_maps):
    """Merge an ordered index group into one (extent, stride per operand) or None."""
    letters = [c for c in letters if size[c] != 1]
    if not letters:
        return 1, tuple(0 for _ in stride_maps)
    for sm in stride_maps:
        for x, y in zip(letters[:-1], letters[1:]):
            if sm[x] != sm[y] * size[y]:
                return None
    ext = 1
    for c in letters:
        ext *= size[c]
    return ext, tuple(sm[letters[-1]] for sm in stride_maps)


def contract(spec: str, A: DevArray, B: DevArray, out: Optional[DevArray] = None, alpha: float = 1.0,
             accumulate: bool = False, k_scale: Optional[DevArray] = None, stream: int = 0,
             split_k: int = 0, _depth: int = 0) -> DevArray:
    """Two-operand einsum ``"<A idx>,<B idx>-><C idx>"`` on the device."""
    lhs, co = spec.replace(" ", "").split("->")
    ia, ib = lhs.split(",")
    if len(ia) != A.ndim or len(ib) != B.ndim:
        raise ValueError(f"contract {spec}: operand ranks {A.ndim}, {B.ndim}")
    size = {}
    for idx, arr in ((ia, A), (ib, B)):
        for c, n in zip(idx, arr.shape):
            if size.setdefault(c, n) != n:
                raise ValueError(f"contract {spec}: extent mismatch on '{c}'")
    out_shape = tuple(size[c] for c in co)
    if out is None:
        out = DevArray.empty(out_shape, stream=stream)
        accumulate = False
    elif out.shape != out_shape:
        raise ValueError(f"contract {spec}: out has shape {out.shape}, expected {out_shape}")
    if out.size == 0:
        return out
    sa = dict(zip(ia, A.strides))
    sb = dict(zip(ib, B.strides))
    sc = dict(zip(co, out.strides))
    batch = [c for c in co if c in sa and c in sb]
    Mg = [c for c in co if c in sa and c not in sb]
    Ng = [c for c in co if c in sb and c not in sa]
    Kg = [c for c in ia if c in sb and c not in sc]
    extra = [c for c in ia + ib if c not in co and not (c in sa and c in sb)]
    if extra:
        raise ValueError(f"contract {spec}: index '{extra[0]}' is summed in one operand only")

    # Letters of the M (or N) group that do not merge with the rest can ride in the batch
    # dimension with stride 0 on the operand that lacks them (a batched product).
    sa0 = {c: sa.get(c, 0) for c in size}
    sb0 = {c: sb.get(c, 0) for c in size}
    mb = mm = mn = None
    for tm in range(len(Mg) + 1):
        for tn in range(len(Ng) + 1):
            if tm and tn:
                continue
            bgroup = [c for c in co if c in batch or c in Mg[:tm] or c in Ng[:tn]]
            mb = _merge(bgroup, size, sa0, sb0, sc)
            mm = _merge(Mg[tm:], size, sa, sc)
            mn = _merge(Ng[tn:], size, sb, sc)
            if mb is not None and mm is not None and mn is not None:
                break
        if mb is not None and mm is not None and mn is not None:
            break
    # K: try single merged dim (any order), else two dims
    Kn = [c for c in Kg if size[c] != 1]
    kplan = None
    import itertools
    for perm in itertools.permutations(Kn):
        m = _merge(list(perm), size, sa, sb)
        if m is not None:
            kplan = (1, 0, 0, m[0], m[1][0], m[1][1])
            break
    if kplan is None:
        for perm in itertools.permutations(Kn):
            for cut in range(1, len(perm)):
                m1 = _merge(list(perm[:cut]), size, sa, sb)
                m2 = _merge(list(perm[cut:]), size, sa, sb)
                if m1 is not None and m2 is not None:
                    kplan = (m1[0], m1[1][0], m1[1][1], m2[0], m2[1][0], m2[1][1])
                    break
            if kplan:
                break
    if mb is None or mm is None or mn is None or kplan is None:
        # materialise the offending operand(s) in [batch, M, K] / [batch, K, N] order and retry
        if _depth > 0:
            raise ValueError(f"contract {spec}: cannot be lowered onto the strided GEMM")
        if not out.is_contiguous():
            raise ValueError(f"contract {spec}: a strided `out` needs mergeable index groups")
        oa, ob = batch + Mg + Kg, batch + Kg + Ng
        A2 = A.transpose([ia.index(c) for c in oa]).contiguous(stream)
        B2 = B.transpose([ib.index(c) for c in ob]).contiguous(stream)
        return contract(f"{''.join(oa)},{''.join(ob)}->{co}", A2, B2, out, alpha, accumulate, k_scale,
                        stream, split_k, _depth=1)

    d = nat.GemmDesc()
    d.batch, (d.a_b, d.b_b, d.c_b) = mb[0], mb[1]
    d.M, (d.a_m, d.c_m) = mm[0], mm[1]
    d.N, (d.b_n, d.c_n) = mn[0], mn[1]
    d.Ko, d.a_ko, d.b_ko, d.Ki, d.a_ki, d.b_ki = kplan
    d.alpha = float(alpha)
    d.accumulate = 1 if accumulate else 0
    d.split_k = int(split_k)
    ks = None
    if k_scale is not None:
        if not k_scale.is_contiguous() or k_scale.size != d.Ko * d.Ki:
            raise ValueError("k_scale must be a contiguous vector over the contracted index")
        ks = ctypes.c_void_p(k_scale.ptr)
    nat.call("ttsk_gemm", ctypes.byref(d), ctypes.c_void_p(A.ptr), ctypes.c_void_p(B.ptr),
             ctypes.c_void_p(out.ptr), ks, stream)
    return out


def axpby(y: DevArray, x: DevArray, a: float = 1.0, b: float = 1.0, stream=0):
    """y <- a*x + b*y (contiguous, same shape)."""
    if y.shape != x.shape:
        raise ValueError("axpby: shape mismatch")
    x = x.contiguous(stream)
    if not y.is_contiguous():
        raise ValueError("axpby: destination must be contiguous")
    if y.size:
        nat.call("ttsk_axpby", ctypes.c_void_p(y.ptr), ctypes.c_void_p(x.ptr), float(a), float(b),
                 ctypes.c_size_t(y.size), stream)
    return y
