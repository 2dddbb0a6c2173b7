"""ctypes binding of libblsq_hip.so (include/blsq.h) — no torch, no fallback.

The library is built in-tree by ``bounded-lsq_amd/csrc/Makefile`` (see
``__graft_entry__.build``).  If it is missing, or no GPU is visible when a
context is requested, this module raises: there is deliberately NO CPU path
behind the product API.
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BLSQ_LIB", os.path.join(_HERE, "libblsq_hip.so"))

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)
vp = C.c_void_p

# every symbol include/blsq.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "blsq_version": (C.c_int, []),
    "blsq_device_count": (C.c_int, []),
    "blsq_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "blsq_ctx_destroy": (C.c_int, [vp]),
    "blsq_last_error": (C.c_char_p, [vp]),
    "blsq_sync": (C.c_int, [vp]),
    "blsq_dev_malloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "blsq_dev_free": (C.c_int, [vp, vp]),
    "blsq_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "blsq_host_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "blsq_host_free": (C.c_int, [vp, vp]),
    "blsq_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "blsq_timing_enable": (C.c_int, [vp, C.c_int]),
    "blsq_timing_reset": (C.c_int, [vp]),
    "blsq_timing_count": (C.c_int, [vp]),
    "blsq_timing_get": (C.c_int, [vp, C.c_int, C.POINTER(C.c_char_p), c_double_p, c_int64_p]),
    "blsq_trf_plan_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "blsq_trf_plan_destroy": (C.c_int, [vp]),
    "blsq_trf_factor": (C.c_int, [vp] + [vp] * 6 + [C.c_int] + [vp] * 3),
    "blsq_trf_step": (C.c_int, [vp, vp, vp, C.c_double] + [vp] * 11),
    "blsq_trf_factor_dev": (C.c_int, [vp] + [vp] * 6 + [C.c_int]),
    "blsq_trf_step_dev": (C.c_int, [vp, vp, vp, C.c_double]),
    "blsq_trf_fetch_factor": (C.c_int, [vp] + [vp] * 5),
    "blsq_trf_fetch_step": (C.c_int, [vp] + [vp] * 15),
    "blsq_debug_qr_stamps": (C.c_int, [vp]),
    "blsq_trf_debug_fast": (C.c_int, [vp, vp]),
    "blsq_trf_debug_sweeps": (C.c_int, [vp, vp]),
    "blsq_trf_debug_cond": (C.c_int, [vp, vp]),
    "blsq_dogbox_debug_cond": (C.c_int, [vp, vp]),
    "blsq_dogbox_plan_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "blsq_dogbox_plan_destroy": (C.c_int, [vp]),
    "blsq_dogbox_factor": (C.c_int, [vp] + [vp] * 6 + [C.c_int] + [vp] * 5),
    "blsq_dogbox_step": (C.c_int, [vp] + [vp] * 9),
    "blsq_dogbox_factor_dev": (C.c_int, [vp] + [vp] * 6 + [C.c_int, vp]),
    "blsq_dogbox_step_dev": (C.c_int, [vp, vp]),
    "blsq_dogbox_fetch_factor": (C.c_int, [vp] + [vp] * 7),
    "blsq_dogbox_fetch_step": (C.c_int, [vp] + [vp] * 8),
    "blsq_tsqr_tri_ld": (C.c_int, [C.c_int]),
    "blsq_comm_id_bytes": (C.c_int, []),
    "blsq_comm_get_id": (C.c_int, [vp, vp, C.c_size_t]),
    "blsq_comm_init": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_size_t]),
    "blsq_comm_destroy": (C.c_int, [vp]),
    "blsq_comm_library": (C.c_char_p, [c_int32_p]),
    "blsq_comm_size": (C.c_int, [vp]),
    "blsq_comm_rank": (C.c_int, [vp]),
    "blsq_comm_allreduce_max": (C.c_int, [vp, c_double_p, C.c_int]),
    "blsq_tsqr_plan_create": (C.c_int, [vp, C.c_int, C.c_longlong, C.c_int, C.c_int, C.POINTER(vp)]),
    "blsq_tsqr_factor_dev": (C.c_int, [vp] + [vp] * 6 + [C.c_int]),
    "blsq_tsqr_local_dev": (C.c_int, [vp, vp, vp, vp]),
    "blsq_tsqr_combine_dev": (C.c_int, [vp] + [vp] * 5 + [C.c_int]),
    "blsq_debug_cqr_stats": (C.c_int, [vp, C.POINTER(C.c_uint64), C.c_int]),
    "blsq_debug_gram_stats": (C.c_int, [vp, C.POINTER(C.c_uint64), C.c_int]),
    "blsq_debug_cqr2_stats": (C.c_int, [vp, C.POINTER(C.c_uint64), C.c_int]),
    "blsq_option_count": (C.c_int, []),
    "blsq_option_info": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                   C.POINTER(C.c_char_p)]),
    "blsq_ctx_set_option": (C.c_int, [vp, C.c_char_p, C.c_double]),
    "blsq_ctx_get_option": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_double)]),
    "blsq_debug_csne_stats": (C.c_int, [vp, C.POINTER(C.c_uint64), C.c_int]),
    "blsq_trf_debug_csne": (C.c_int, [vp, vp, vp]),
    "blsq_debug_probe": (C.c_int, [vp, C.c_int, C.c_int, c_double_p]),
    "blsq_outer_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "blsq_outer_destroy": (C.c_int, [vp]),
    "blsq_outer_buffers": (C.c_int, [vp] + [C.POINTER(vp)] * 6),
    "blsq_outer_start": (C.c_int, [vp] + [vp] * 5 + [C.c_int, C.c_double, C.c_double, C.c_double,
                                                    C.c_int]),
    "blsq_outer_begin": (C.c_int, [vp]),
    "blsq_outer_propose": (C.c_int, [vp, c_int32_p]),
    "blsq_outer_judge": (C.c_int, [vp, c_int32_p]),
    "blsq_outer_fetch": (C.c_int, [vp] + [vp] * 8),
    "blsq_fd_points_dev": (C.c_int, [vp, C.c_int, C.c_int, C.c_int] + [vp] * 7),
    "blsq_fd_assemble_dev": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int] + [vp] * 7),
}

_lib = None


class BlsqError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and attach the declared signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BlsqError(
            "libblsq_hip.so is not built (%s). Run `python -c 'import "
            "__graft_entry__ as g; g.build()'` or `make -C bounded-lsq_amd/csrc`. "
            "There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(a):
    """void* of a numpy array (or None)."""
    if a is None:
        return None
    return a.ctypes.data_as(vp)


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError("expected shape %r, got %r" % (tuple(shape), a.shape))
    return a


class Context:
    """One device + one HIP stream (blsq_ctx).  Not thread-safe."""

    def __init__(self, device_id=0):
        self.lib = load()
        h = vp()
        rc = self.lib.blsq_ctx_create(int(device_id), C.byref(h))
        if rc != 0:
            raise BlsqError("blsq_ctx_create(device=%d) failed with code %d "
                            "(no MI355X visible?)" % (device_id, rc))
        self.h = h
        self.device_id = int(device_id)
        self._plans = weakref.WeakSet()      # plans / drivers created on this ctx: closed before it

    def adopt(self, plan):
        """Register an object whose close() must run before the context is destroyed (the
        library's plans keep a pointer to their ctx)."""
        self._plans.add(plan)

    def check(self, rc, what=""):
        if rc != 0:
            msg = self.lib.blsq_last_error(self.h)
            raise BlsqError("%s failed (%d): %s" % (what, rc, (msg or b"").decode()))

    def close(self):
        """Close every adopted plan / driver, then destroy the context — exactly once: the plan set
        is taken first, so a plan that owns this context and closes it from its own close()
        (OuterDriver with ctx=None) finds nothing left to walk."""
        if not getattr(self, "h", None) or getattr(self, "_closing", False):
            return
        self._closing = True
        plans, self._plans = list(getattr(self, "_plans", ())), weakref.WeakSet()
        try:
            for pl in plans:                     # (plans destroy their handles through the live ctx)
                try:
                    pl.close()
                except (BlsqError, OSError, AttributeError):
                    pass
        finally:
            h, self.h = self.h, None
            self._closing = False
            self.lib.blsq_ctx_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self.check(self.lib.blsq_sync(self.h), "blsq_sync")

    # ---- device memory -----------------------------------------------------
    def malloc(self, nbytes):
        p = vp()
        self.check(self.lib.blsq_dev_malloc(self.h, int(nbytes), C.byref(p)), "blsq_dev_malloc")
        return p

    def free(self, p):
        self.check(self.lib.blsq_dev_free(self.h, p), "blsq_dev_free")

    def pinned_empty(self, shape, dtype=np.float64):
        """A numpy array in page-locked host memory (blsq_host_alloc): hand it to factor() / let `jac` fill it and
        the Jacobian crosses PCIe by DMA at link speed.  Keep the context alive as long as the array; release with
        ``ctx.pinned_free(arr)`` (or leave it to the process)."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        p = vp()
        self.check(self.lib.blsq_host_alloc(self.h, nbytes, C.byref(p)), "blsq_host_alloc")
        buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.__array_interface__["data"][0]] = p
        return arr

    def pinned_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.__array_interface__["data"][0], None)
        if p is not None:
            self.check(self.lib.blsq_host_free(self.h, p), "blsq_host_free")

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(arr.nbytes)
        self.check(self.lib.blsq_memcpy_h2d(self.h, p, ptr(arr), arr.nbytes), "blsq_memcpy_h2d")
        return p

    def to_host(self, p, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self.check(self.lib.blsq_memcpy_d2h(self.h, ptr(out), p, out.nbytes), "blsq_memcpy_d2h")
        return out

    def cqr_stats(self, reset=False):
        """-> (panels on the Cholesky-QR fast path, panels on the Householder column loop)."""
        out = (C.c_uint64 * 2)()
        self.check(self.lib.blsq_debug_cqr_stats(self.h, out, 1 if reset else 0), "cqr_stats")
        return int(out[0]), int(out[1])

    def gram_stats(self, reset=False):
        """-> (problems factored by the normal-equations fast path, problems handed to the QR tree)."""
        out = (C.c_uint64 * 2)()
        self.check(self.lib.blsq_debug_gram_stats(self.h, out, 1 if reset else 0), "gram_stats")
        return int(out[0]), int(out[1])

    def cqr2_stats(self, reset=False):
        """-> problems (of those the gate handed on) factored by the CholeskyQR2 middle tier."""
        out = (C.c_uint64 * 1)()
        self.check(self.lib.blsq_debug_cqr2_stats(self.h, out, 1 if reset else 0), "cqr2_stats")
        return int(out[0])

    # ---- the option table (csrc/blsq_options.cpp) -----------------------------------------------
    def set_option(self, name, value):
        """One switch of THIS ctx (`name`: the table's key or its environment variable).  Route switches take effect for
        plans created afterwards."""
        self.check(self.lib.blsq_ctx_set_option(self.h, str(name).encode(), float(value)), "blsq_ctx_set_option")

    def get_option(self, name):
        v = C.c_double(0.0)
        rc = self.lib.blsq_ctx_get_option(self.h, str(name).encode(), C.byref(v))
        if rc != 0:
            raise KeyError(name)
        return float(v.value)

    def options(self):
        """-> list of dict(name, env, default, value, doc): every switch of the library and its value on this ctx."""
        out = []
        for i in range(self.lib.blsq_option_count()):
            nm, ev, doc = C.c_char_p(), C.c_char_p(), C.c_char_p()
            df = C.c_double(0.0)
            self.lib.blsq_option_info(i, C.byref(nm), C.byref(ev), C.byref(df), C.byref(doc))
            out.append({"name": nm.value.decode(), "env": ev.value.decode(), "default": float(df.value),
                        "value": self.get_option(nm.value.decode()), "doc": doc.value.decode()})
        return out

    def csne_stats(self, reset=False):
        """-> (problems routed to the CSNE tier by factor calls, step-solves it delivered, step-solves it declined)."""
        out = (C.c_uint64 * 3)()
        self.check(self.lib.blsq_debug_csne_stats(self.h, out, 1 if reset else 0), "csne_stats")
        return int(out[0]), int(out[1]), int(out[2])

    # ---- collective over the ranks of one tall problem (RCCL, blsq_comm_*) ---------------------
    def comm_new_id(self):
        """Rendezvous id (bytes) — call on ONE rank and hand it to the others."""
        n = self.lib.blsq_comm_id_bytes()
        buf = (C.c_char * n)()
        self.check(self.lib.blsq_comm_get_id(self.h, buf, n), "blsq_comm_get_id")
        return bytes(buf)

    def comm_init(self, nranks, rank, comm_id):
        buf = C.create_string_buffer(bytes(comm_id), len(comm_id))
        self.check(self.lib.blsq_comm_init(self.h, int(nranks), int(rank), buf, len(comm_id)),
                   "blsq_comm_init")

    def comm_info(self):
        """-> dict(library=<path of the collective library in use>, version=<ncclGetVersion code>,
        ranks=<size of this ctx's communicator>, rank=...)"""
        v = C.c_int32(0)
        path = self.lib.blsq_comm_library(C.byref(v))
        return {"library": (path or b"").decode(), "version": int(v.value),
                "ranks": int(self.lib.blsq_comm_size(self.h)), "rank": int(self.lib.blsq_comm_rank(self.h))}

    def comm_destroy(self):
        self.check(self.lib.blsq_comm_destroy(self.h), "blsq_comm_destroy")

    def comm_max(self, values):
        """Max over the ranks (also a barrier); `values`: up to 64 floats."""
        arr = np.ascontiguousarray(values, dtype=np.float64).copy()
        self.check(self.lib.blsq_comm_allreduce_max(self.h, arr.ctypes.data_as(c_double_p),
                                                    arr.size), "blsq_comm_allreduce_max")
        return arr

    def probe(self, kind, arg):
        """Measured peaks (blsq_debug_probe): kind 'mfma_f64' (arg = waves per SIMD) -> (TFLOP/s,
        MFMA wave-instructions, ms); kind 'copy' (arg = MiB) -> (GB/s read + written, bytes, ms)."""
        out = (C.c_double * 3)()
        self.check(self.lib.blsq_debug_probe(self.h, {"mfma_f64": 0, "copy": 1}[kind], int(arg), out),
                   "blsq_debug_probe")
        return float(out[0]), float(out[1]), float(out[2])

    # ---- timing ------------------------------------------------------------
    def timing(self, on=True, only=None):
        """Per-kernel HIP-event timing: every slot, or (only = slot name) the launches of one slot — two events
        around EVERY launch cost the step about 2 %."""
        mode = 1 if on else 0
        if on and only is not None:
            names = []
            for i in range(self.lib.blsq_timing_count(self.h)):
                name = C.c_char_p(); ms = C.c_double(); cnt = C.c_int64()
                self.check(self.lib.blsq_timing_get(self.h, i, C.byref(name), C.byref(ms), C.byref(cnt)), "timing_get")
                names.append(name.value.decode())
            mode = 2 + names.index(only)
        self.check(self.lib.blsq_timing_enable(self.h, mode), "timing_enable")

    def timing_reset(self):
        self.check(self.lib.blsq_timing_reset(self.h), "timing_reset")

    def timing_read(self):
        """-> {kernel slot name: (total_ms, launches)} (call after sync)."""
        out = {}
        for i in range(self.lib.blsq_timing_count(self.h)):
            name = C.c_char_p()
            ms = C.c_double()
            cnt = C.c_int64()
            self.check(self.lib.blsq_timing_get(self.h, i, C.byref(name), C.byref(ms),
                                                C.byref(cnt)), "timing_get")
            out[name.value.decode()] = (ms.value, cnt.value)
        return out
