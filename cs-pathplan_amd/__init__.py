"""cs-pathplan_amd -- MI355X-native batched minimum-snap solver (Python binding of the C-ABI).

The directory name carries a hyphen, so import it with

    import importlib; csp = importlib.import_module("cs-pathplan_amd")

This module is a thin ctypes layer over libcsp_minsnap.so (include/csp_minsnap.h).  It is used
by tests/, bench.py and the torch.distributed sharding helper; the drop-in for the reference's
C++ callers is the class shim in host/math_util/minimum_snap.hpp.

There is NO CPU fallback: importing fails loudly when the HIP extension has not been built,
and every solve call raises when no gfx950 device is visible.
"""
import ctypes
import os

import numpy as np

try:
    # Must precede the CDLL below.  torch bundles its own libamdhip64.so.7; if ours (linked to
    # /opt/rocm) were loaded first the process would hold two HIP runtimes and torch would then
    # report "No HIP GPUs are available".  Loading torch first makes both share one runtime.
    import torch  # noqa: F401
except ImportError:  # the C-ABI itself does not need torch (host-memory calls, C++ callers)
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcsp_minsnap.so")

ABI_VERSION = 1
DTYPE_F64, DTYPE_F32 = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
FLAG_FORCE_GENERIC = 0x1
FLAG_SEGMENT_MAJOR = 0x2
FLAG_NO_PERSISTENT = 0x4
FLAG_F32_ARITH = 0x8
FLAG_LONG_SEGMENTS = 0x10
FLAG_SPAN = 0x20
TRAJ_OK, TRAJ_NONFINITE, TRAJ_NOT_SPD, TRAJ_SKIPPED = 0, 1, 2, 4

EXPORTED_SYMBOLS = (
    "csp_minsnap_solve_batch", "csp_minsnap_solve_batch_sharded", "csp_minsnap_workspace_bytes", "csp_minsnap_time_alloc_batch",
    "csp_minsnap_solve_mixed", "csp_minsnap_mixed_workspace_bytes", "csp_minsnap_solve_multi",
    "csp_minsnap_plan_batch", "csp_minsnap_plan_workspace_bytes", "csp_minsnap_sample_batch",
    "csp_minsnap_generate_batch", "csp_minsnap_sample_capacity",
    "csp_minsnap_kernel_name", "csp_minsnap_device_count", "csp_minsnap_version",
    "csp_minsnap_strerror", "csp_minsnap_last_hip_error", "csp_minsnap_release_cached_memory",
    "csp_geo_wgs84_to_enu_batch", "csp_geo_enu_to_wgs84_batch",
    "csp_alt_workspace_bytes", "csp_alt_optimize_heights_batch", "csp_alt_global_smooth_batch",
    "csp_bezier_generate_batch",
)


class CspError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        super().__init__("csp_minsnap error %d (%s)%s" % (code, strerror(code), (": " + detail) if detail else ""))


class Desc(ctypes.Structure):
    """Mirror of `csp_minsnap_desc` (include/csp_minsnap.h)."""
    _fields_ = [
        ("abi_version", ctypes.c_uint32), ("dtype", ctypes.c_uint32),
        ("order", ctypes.c_int32), ("num_segments", ctypes.c_int32),
        ("batch", ctypes.c_int64),
        ("seg_offsets", ctypes.c_void_p),
        ("max_segments", ctypes.c_int32), ("bc_per_trajectory", ctypes.c_uint32),
        ("path_weight", ctypes.c_double), ("vel_zero_weight", ctypes.c_double),
        ("vel_zero_weight_per_traj", ctypes.c_void_p),
        ("mem_space", ctypes.c_uint32), ("device_id", ctypes.c_int32),
        ("flags", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
    ]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "cs-pathplan_amd: %s is missing.  Build the HIP extension first "
        "(python cs-pathplan_amd/build.py, or __graft_entry__.build()); there is no CPU fallback." % LIB_PATH)

_lib = ctypes.CDLL(LIB_PATH)
_lib.csp_minsnap_solve_batch.restype = ctypes.c_int
_lib.csp_minsnap_solve_batch.argtypes = [ctypes.POINTER(Desc)] + [ctypes.c_void_p] * 7 + [ctypes.c_size_t, ctypes.c_void_p]
_lib.csp_minsnap_solve_batch_sharded.restype = ctypes.c_int
_lib.csp_minsnap_solve_batch_sharded.argtypes = [ctypes.POINTER(Desc)] + [ctypes.c_void_p] * 6 + [ctypes.c_int]
_lib.csp_minsnap_workspace_bytes.restype = ctypes.c_size_t
_lib.csp_minsnap_workspace_bytes.argtypes = [ctypes.POINTER(Desc)]
_lib.csp_minsnap_solve_multi.restype = ctypes.c_int
_lib.csp_minsnap_solve_multi.argtypes = [ctypes.POINTER(Desc), ctypes.c_int] + [ctypes.c_void_p] * 7
_lib.csp_minsnap_solve_mixed.restype = ctypes.c_int
_lib.csp_minsnap_solve_mixed.argtypes = [ctypes.POINTER(Desc)] + [ctypes.c_void_p] * 8 + [ctypes.c_size_t, ctypes.c_void_p]
_lib.csp_minsnap_mixed_workspace_bytes.restype = ctypes.c_size_t
_lib.csp_minsnap_mixed_workspace_bytes.argtypes = [ctypes.POINTER(Desc)]
_lib.csp_minsnap_time_alloc_batch.restype = ctypes.c_int
_lib.csp_minsnap_time_alloc_batch.argtypes = [ctypes.POINTER(Desc), ctypes.c_void_p, ctypes.c_double,
                                              ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
_lib.csp_minsnap_plan_batch.restype = ctypes.c_int
_lib.csp_minsnap_plan_batch.argtypes = [ctypes.POINTER(Desc), ctypes.c_void_p, ctypes.c_double, ctypes.c_double] + \
    [ctypes.c_void_p] * 8 + [ctypes.c_size_t, ctypes.c_void_p]
_lib.csp_minsnap_plan_workspace_bytes.restype = ctypes.c_size_t
_lib.csp_minsnap_plan_workspace_bytes.argtypes = [ctypes.POINTER(Desc)]
_lib.csp_minsnap_sample_batch.restype = ctypes.c_int
_lib.csp_minsnap_sample_batch.argtypes = [ctypes.POINTER(Desc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double,
                                          ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
_lib.csp_minsnap_generate_batch.restype = ctypes.c_int
_lib.csp_minsnap_generate_batch.argtypes = [ctypes.POINTER(Desc), ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_void_p,
                                            ctypes.c_double, ctypes.c_int64] + [ctypes.c_void_p] * 10 + [ctypes.c_size_t, ctypes.c_void_p]
_lib.csp_minsnap_sample_capacity.restype = ctypes.c_int64
_lib.csp_minsnap_sample_capacity.argtypes = [ctypes.POINTER(Desc), ctypes.c_void_p, ctypes.c_double, ctypes.c_double]
for _n in ("csp_geo_wgs84_to_enu_batch", "csp_geo_enu_to_wgs84_batch"):
    getattr(_lib, _n).restype = ctypes.c_int
    getattr(_lib, _n).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint32,
                                  ctypes.c_int32, ctypes.c_void_p]
_lib.csp_minsnap_kernel_name.restype = ctypes.c_char_p
_lib.csp_minsnap_kernel_name.argtypes = [ctypes.POINTER(Desc)]
_lib.csp_minsnap_device_count.restype = ctypes.c_int
_lib.csp_minsnap_version.restype = ctypes.c_char_p
_lib.csp_minsnap_strerror.restype = ctypes.c_char_p
_lib.csp_minsnap_strerror.argtypes = [ctypes.c_int]
_lib.csp_minsnap_last_hip_error.restype = ctypes.c_char_p


_lib.csp_minsnap_release_cached_memory.restype = None


def release_cached_memory():
    """Frees the idle staging arenas host-memory calls keep between calls (include/csp_minsnap.h)."""
    _lib.csp_minsnap_release_cached_memory()


def raw_lib():
    return _lib


def version():
    return _lib.csp_minsnap_version().decode()


def strerror(code):
    return _lib.csp_minsnap_strerror(int(code)).decode()


def device_count():
    return int(_lib.csp_minsnap_device_count())


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _np_dtype(dtype_code):
    return np.float32 if dtype_code == DTYPE_F32 else np.float64


def make_desc(order, batch, num_segments=0, dtype=DTYPE_F64, path_weight=0.0, vel_zero_weight=0.0,
              mem_space=MEM_HOST, bc_per_trajectory=False, seg_offsets_ptr=None, max_segments=0,
              vw_per_ptr=None, device_id=-1, flags=0):
    d = Desc()
    d.abi_version = ABI_VERSION
    d.dtype = dtype
    d.order = int(order)
    d.num_segments = int(num_segments)
    d.batch = int(batch)
    d.seg_offsets = seg_offsets_ptr
    d.max_segments = int(max_segments)
    d.bc_per_trajectory = 1 if bc_per_trajectory else 0
    d.path_weight = float(path_weight)
    d.vel_zero_weight = float(vel_zero_weight)
    d.vel_zero_weight_per_traj = vw_per_ptr
    d.mem_space = mem_space
    d.device_id = int(device_id)
    d.flags = int(flags)
    d.reserved = 0
    return d


def workspace_bytes(desc):
    return int(_lib.csp_minsnap_workspace_bytes(ctypes.byref(desc)))


def kernel_name(desc):
    r = _lib.csp_minsnap_kernel_name(ctypes.byref(desc))
    return r.decode() if r else None


def _check(rc):
    if rc != 0:
        raise CspError(rc, _lib.csp_minsnap_last_hip_error().decode() if rc == -4 or rc == -5 else "")


class Result:
    __slots__ = ("coeffs", "max_dev", "status", "kernel")

    def __init__(self, coeffs, max_dev, status, kernel):
        self.coeffs, self.max_dev, self.status, self.kernel = coeffs, max_dev, status, kernel


def solve_batch(waypoints, times, bc=None, order=4, path_weight=0.0, vel_zero_weight=0.0,
                seg_offsets=None, max_segments=None, vel_zero_weight_per_traj=None,
                want_max_dev=False, want_status=False, out=None, workspace=None, stream=None, ngpu=None,
                force_generic=False, segment_major=False, no_persistent=False, f32_arith=False, span=False):
    """Batched SolveQPClosedForm (math_util/minimum_snap.hpp:45-53).

    numpy inputs  -> CSP_MEM_HOST (staged through the device, synchronous);
    torch CUDA tensors -> CSP_MEM_DEVICE (enqueued on `stream` or torch's current stream).
    Uniform batches: waypoints [B,S+1,3], times [B,S]; ragged: pass seg_offsets [B+1] (int64)
    with concatenated waypoints [sum(S_b+1),3] and times [sum S_b].
    bc: [4,3] / [1,4,3] shared or [B,4,3] per trajectory; rows start vel, end vel, start acc,
    end acc (reference Vel/Acc); None = zeros (MinimumSnapConfig defaults, minimum_snap.hpp:29-32).
    Returns Result(coeffs [B,S,3,2o] (ragged: [sum S_b,3,2o]), max_dev, status, kernel name).
    """
    on_device = _is_torch(waypoints)
    ragged = seg_offsets is not None
    flags = ((FLAG_FORCE_GENERIC if force_generic else 0) | (FLAG_SEGMENT_MAJOR if segment_major else 0)
             | (FLAG_NO_PERSISTENT if no_persistent else 0) | (FLAG_F32_ARITH if f32_arith else 0)
             | (FLAG_SPAN if span else 0))
    if segment_major and ragged:
        raise ValueError("segment_major needs a uniform batch")
    m = 2 * int(order)
    if on_device:
        import torch
        if not waypoints.is_cuda:
            raise ValueError("torch inputs must be CUDA tensors (use numpy arrays for host memory)")
        tdt = waypoints.dtype
        dtype = DTYPE_F32 if tdt == torch.float32 else DTYPE_F64
        dev = waypoints.device
        waypoints, times = waypoints.contiguous(), times.to(tdt).contiguous()
        if ragged:
            seg_offsets = seg_offsets.to(device=dev, dtype=torch.int64).contiguous()
            B = seg_offsets.numel() - 1
            total = times.numel()
            if max_segments is None:
                max_segments = int((seg_offsets[1:] - seg_offsets[:-1]).max().item()) if B else 1
            S = 0
        else:
            B, S = times.shape
            total = B * S
        if bc is None:
            bc = torch.zeros((1, 4, 3), dtype=tdt, device=dev)
        bc = bc.to(tdt).contiguous().reshape(-1, 4, 3)
        if bc.shape[0] not in (1, B):
            raise ValueError("bc must be [4,3], [1,4,3] or [B,4,3]")
        per = bc.shape[0] == B
        if out is None:
            out = torch.empty((total, 3, m) if ragged else ((S, B, 3, m) if segment_major else (B, S, 3, m)), dtype=tdt, device=dev)
        md = torch.empty(B, dtype=torch.float64, device=dev) if want_max_dev else None
        stt = torch.empty(B, dtype=torch.int32, device=dev) if want_status else None
        vwp = None
        if vel_zero_weight_per_traj is not None:
            vwp = vel_zero_weight_per_traj.to(device=dev, dtype=torch.float64).contiguous()
        desc = make_desc(order, B, S, dtype, path_weight, vel_zero_weight, MEM_DEVICE, per,
                         seg_offsets.data_ptr() if ragged else None, max_segments or 0,
                         vwp.data_ptr() if vwp is not None else None,
                         dev.index if dev.index is not None else -1, flags)
        need = workspace_bytes(desc)
        if need and (workspace is None or workspace.numel() * workspace.element_size() < need):
            workspace = torch.empty(need, dtype=torch.uint8, device=dev)
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        if ngpu is not None:
            # the batch is resident on THIS (root) device: scatter / solve / gather over RCCL from one process, synchronous
            # (include/csp_minsnap.h: csp_minsnap_solve_batch_sharded with CSP_MEM_DEVICE)
            rc = _lib.csp_minsnap_solve_batch_sharded(
                ctypes.byref(desc), waypoints.data_ptr(), times.data_ptr(), bc.data_ptr(), out.data_ptr(),
                md.data_ptr() if md is not None else None, stt.data_ptr() if stt is not None else None, int(ngpu))
            _check(rc)
            return Result(out, md, stt, kernel_name(desc))
        rc = _lib.csp_minsnap_solve_batch(
            ctypes.byref(desc), waypoints.data_ptr(), times.data_ptr(), bc.data_ptr(), out.data_ptr(),
            md.data_ptr() if md is not None else None, stt.data_ptr() if stt is not None else None,
            workspace.data_ptr() if need else None, need, ctypes.c_void_p(st))
        _check(rc)
        return Result(out, md, stt, kernel_name(desc))

    # host memory
    waypoints = np.asarray(waypoints)
    dtype = DTYPE_F32 if waypoints.dtype == np.float32 else DTYPE_F64
    npdt = _np_dtype(dtype)
    waypoints = np.ascontiguousarray(waypoints, dtype=npdt)
    times = np.ascontiguousarray(times, dtype=npdt)
    if ragged:
        seg_offsets = np.ascontiguousarray(seg_offsets, dtype=np.int64)
        B = seg_offsets.shape[0] - 1
        total = times.shape[0]
        if max_segments is None:
            max_segments = int(np.max(np.diff(seg_offsets))) if B else 1
        S = 0
    else:
        B, S = times.shape
        total = B * S
    bc = np.zeros((1, 4, 3), dtype=npdt) if bc is None else np.ascontiguousarray(bc, dtype=npdt).reshape(-1, 4, 3)
    if bc.shape[0] not in (1, B):
        raise ValueError("bc must be [4,3], [1,4,3] or [B,4,3]")
    per = bc.shape[0] == B
    if out is None:
        out = np.empty((total, 3, m) if ragged else ((S, B, 3, m) if segment_major else (B, S, 3, m)), dtype=npdt)
    md = np.empty(B, dtype=np.float64) if want_max_dev else None
    stt = np.empty(B, dtype=np.int32) if want_status else None
    vwp = None
    if vel_zero_weight_per_traj is not None:
        vwp = np.ascontiguousarray(vel_zero_weight_per_traj, dtype=np.float64)
    desc = make_desc(order, B, S, dtype, path_weight, vel_zero_weight, MEM_HOST, per,
                     seg_offsets.ctypes.data if ragged else None, max_segments or 0,
                     vwp.ctypes.data if vwp is not None else None, -1, flags)
    if ngpu is not None:   # one process, the batch cut into contiguous chunks over `ngpu` devices (host arrays only)
        rc = _lib.csp_minsnap_solve_batch_sharded(
            ctypes.byref(desc), waypoints.ctypes.data, times.ctypes.data, bc.ctypes.data, out.ctypes.data,
            md.ctypes.data if md is not None else None, stt.ctypes.data if stt is not None else None, int(ngpu))
    else:
        rc = _lib.csp_minsnap_solve_batch(
            ctypes.byref(desc), waypoints.ctypes.data, times.ctypes.data, bc.ctypes.data, out.ctypes.data,
            md.ctypes.data if md is not None else None, stt.ctypes.data if stt is not None else None,
            None, 0, None)
    _check(rc)
    return Result(out, md, stt, kernel_name(desc))


class PreparedMulti:
    """csp_minsnap_solve_multi with everything fixed: `run()` is one C-ABI call -- and one kernel launch for the fixed-size
    buckets -- over n independent uniform batches of one shape (lists of CUDA tensors [B_k,S+1,3] / [B_k,S])."""

    def __init__(self, waypoints, times, bcs=None, order=4, vel_zero_weight=0.0, want_status=False, stream=None):
        import torch
        n = len(waypoints)
        self.dev, tdt = waypoints[0].device, waypoints[0].dtype
        S = times[0].shape[1]
        m = 2 * int(order)
        self.wp = [w.contiguous() for w in waypoints]
        self.tm = [t.to(tdt).contiguous() for t in times]
        zero = torch.zeros((1, 4, 3), dtype=tdt, device=self.dev)
        self.bc = [zero if (bcs is None or bcs[k] is None) else bcs[k].to(tdt).contiguous().reshape(-1, 4, 3) for k in range(n)]
        per = self.bc[0].shape[0] != 1
        self.out = [torch.empty((t.shape[0], S, 3, m), dtype=tdt, device=self.dev) for t in self.tm]
        self.status = [torch.empty(t.shape[0], dtype=torch.int32, device=self.dev) for t in self.tm] if want_status else None
        self.desc = make_desc(order, 0, S, DTYPE_F32 if tdt == torch.float32 else DTYPE_F64, 0.0, vel_zero_weight, MEM_DEVICE, per,
                              device_id=self.dev.index if self.dev.index is not None else -1)
        arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        self._keep = (arr(self.wp), arr(self.tm), arr(self.bc), arr(self.out), arr(self.status) if want_status else None,
                      (ctypes.c_int64 * n)(*[t.shape[0] for t in self.tm]))
        self.n, self._stream = n, stream

    def run(self, stream=None):
        import torch
        st = stream if stream is not None else (self._stream if self._stream is not None
                                                else torch.cuda.current_stream(self.dev).cuda_stream)
        wp, tm, bc, out, stt, nb = self._keep
        rc = _lib.csp_minsnap_solve_multi(ctypes.byref(self.desc), self.n, ctypes.cast(nb, ctypes.c_void_p), ctypes.cast(wp, ctypes.c_void_p),
                                          ctypes.cast(tm, ctypes.c_void_p), ctypes.cast(bc, ctypes.c_void_p), ctypes.cast(out, ctypes.c_void_p),
                                          ctypes.cast(stt, ctypes.c_void_p) if stt is not None else None, ctypes.c_void_p(st))
        if rc:
            _check(rc)
        return self.out


class MixedResult:
    """coeffs: flat storage-dtype array, trajectory b's [S_b,3,2*order_b] block at coeff_offsets[b] .. coeff_offsets[b+1]."""
    def __init__(self, coeffs, coeff_offsets, status):
        self.coeffs, self.coeff_offsets, self.status = coeffs, coeff_offsets, status


def mixed_block_elements(orders, seg_offsets, f32):
    """Elements of every trajectory's coefficient block in csp_minsnap_solve_mixed's layout: 6 * order * S rounded up to whole
    16-byte pieces (fp32: a multiple of 4; fp64: no padding).  Host arrays or device tensors; returns the same kind."""
    pad = 4 if f32 else 2
    if _is_torch(seg_offsets):
        e = (seg_offsets[1:] - seg_offsets[:-1]) * 6 * orders.to(seg_offsets.dtype)
        return (e + pad - 1) // pad * pad
    e = np.diff(np.asarray(seg_offsets, dtype=np.int64)) * 6 * np.asarray(orders, dtype=np.int64)
    return (e + pad - 1) // pad * pad


def mixed_coeff_total(orders, seg_offsets, f32=False):
    """Elements of the coefficient array of a mixed batch (host or device shapes)."""
    e = mixed_block_elements(orders, seg_offsets, f32)
    return int(e.sum().item()) if _is_torch(seg_offsets) else int(np.sum(e))


class PreparedMixed:
    """csp_minsnap_solve_mixed with descriptor, buffers and workspace fixed: `run()` is ONE C-ABI call that buckets the batch
    by (order, length class) on the device and solves it, coefficients in the caller's order (include/csp_minsnap.h)."""

    def __init__(self, orders, waypoints, times, seg_offsets, bc=None, vel_zero_weight=0.0, max_segments=None, out=None,
                 want_status=False, stream=None):
        import torch
        if not (_is_torch(waypoints) and waypoints.is_cuda):
            raise ValueError("PreparedMixed takes CUDA tensors (device memory space)")
        self.dev, tdt = waypoints.device, waypoints.dtype
        dtype = DTYPE_F32 if tdt == torch.float32 else DTYPE_F64
        self.wp, self.tm = waypoints.contiguous(), times.to(tdt).contiguous()
        self.off = seg_offsets.to(device=self.dev, dtype=torch.int64).contiguous()
        self.orders = orders.to(device=self.dev, dtype=torch.int32).contiguous()
        B = self.off.numel() - 1
        if max_segments is None:
            max_segments = int((self.off[1:] - self.off[:-1]).max().item()) if B else 1
        self.bc = (torch.zeros((1, 4, 3), dtype=tdt, device=self.dev) if bc is None else bc.to(tdt).contiguous().reshape(-1, 4, 3))
        self.total = mixed_coeff_total(self.orders, self.off, dtype == DTYPE_F32)
        self.out = out if out is not None else torch.empty(max(self.total, 1), dtype=tdt, device=self.dev)
        self.coeff_offsets = torch.empty(B + 1, dtype=torch.int64, device=self.dev)
        self.status = torch.empty(B, dtype=torch.int32, device=self.dev) if want_status else None
        self.desc = make_desc(0, B, 0, dtype, 0.0, vel_zero_weight, MEM_DEVICE, self.bc.shape[0] == B and B != 1,
                              seg_offsets_ptr=self.off.data_ptr(), max_segments=max_segments,
                              device_id=self.dev.index if self.dev.index is not None else -1, flags=0)
        self.ws_bytes = _lib.csp_minsnap_mixed_workspace_bytes(ctypes.byref(self.desc))
        self.ws = torch.empty(max(self.ws_bytes, 1), dtype=torch.uint8, device=self.dev)
        self._stream = stream
        self._args = (ctypes.byref(self.desc), self.orders.data_ptr(), self.wp.data_ptr(), self.tm.data_ptr(), self.bc.data_ptr(),
                      self.out.data_ptr(), self.coeff_offsets.data_ptr(), self.status.data_ptr() if want_status else None,
                      self.ws.data_ptr(), self.ws_bytes)

    def run(self, stream=None):
        import torch
        st = stream if stream is not None else (self._stream if self._stream is not None
                                                else torch.cuda.current_stream(self.dev).cuda_stream)
        rc = _lib.csp_minsnap_solve_mixed(*self._args, ctypes.c_void_p(st))
        if rc:
            _check(rc)
        return self.out


def solve_mixed(orders, waypoints, times, seg_offsets, bc=None, vel_zero_weight=0.0, vel_zero_weight_per_traj=None,
                max_segments=None, want_status=False, stream=None):
    """csp_minsnap_solve_mixed: a ragged batch whose trajectories carry their own derivative order (2..5).
    numpy inputs -> CSP_MEM_HOST, torch CUDA tensors -> CSP_MEM_DEVICE.  Returns MixedResult."""
    if _is_torch(waypoints):
        import torch
        if vel_zero_weight_per_traj is not None:
            raise ValueError("per-trajectory weights: use the host-memory form or PreparedMixed")
        p = PreparedMixed(orders, waypoints, times, seg_offsets, bc, vel_zero_weight, max_segments, want_status=want_status, stream=stream)
        p.run()
        return MixedResult(p.out, p.coeff_offsets, p.status)
    waypoints = np.asarray(waypoints)
    dtype = DTYPE_F32 if waypoints.dtype == np.float32 else DTYPE_F64
    npdt = _np_dtype(dtype)
    waypoints = np.ascontiguousarray(waypoints, dtype=npdt)
    times = np.ascontiguousarray(times, dtype=npdt)
    seg_offsets = np.ascontiguousarray(seg_offsets, dtype=np.int64)
    orders = np.ascontiguousarray(orders, dtype=np.int32)
    B = seg_offsets.shape[0] - 1
    if max_segments is None:
        max_segments = int(np.max(np.diff(seg_offsets))) if B else 1
    bc = np.zeros((1, 4, 3), dtype=npdt) if bc is None else np.ascontiguousarray(bc, dtype=npdt).reshape(-1, 4, 3)
    out = np.empty(max(mixed_coeff_total(orders, seg_offsets, dtype == DTYPE_F32), 1), dtype=npdt)
    cof = np.empty(B + 1, dtype=np.int64)
    stt = np.empty(B, dtype=np.int32) if want_status else None
    vwp = np.ascontiguousarray(vel_zero_weight_per_traj, dtype=np.float64) if vel_zero_weight_per_traj is not None else None
    desc = make_desc(0, B, 0, dtype, 0.0, vel_zero_weight, MEM_HOST, bc.shape[0] == B and B != 1, seg_offsets.ctypes.data, max_segments,
                     vwp.ctypes.data if vwp is not None else None, -1, 0)
    rc = _lib.csp_minsnap_solve_mixed(ctypes.byref(desc), orders.ctypes.data, waypoints.ctypes.data, times.ctypes.data, bc.ctypes.data,
                                      out.ctypes.data, cof.ctypes.data, stt.ctypes.data if stt is not None else None, None, 0, None)
    _check(rc)
    return MixedResult(out, cof, stt)


class PreparedSolve:
    """A solve whose descriptor, buffers and workspace are fixed: `run()` is one C-ABI call.
    For callers that launch the same shape many times (bench.py, a planner's inner loop) --
    building the descriptor and checking tensors costs more host time than a 40-us kernel."""

    def __init__(self, waypoints, times, bc=None, order=4, path_weight=0.0, vel_zero_weight=0.0, out=None,
                 force_generic=False, segment_major=False, no_persistent=False, stream=None,
                 seg_offsets=None, max_segments=None, span=False):
        import torch
        if not (_is_torch(waypoints) and waypoints.is_cuda):
            raise ValueError("PreparedSolve takes CUDA tensors (device memory space)")
        self.dev, tdt = waypoints.device, waypoints.dtype
        dtype = DTYPE_F32 if tdt == torch.float32 else DTYPE_F64
        self.wp, self.tm = waypoints.contiguous(), times.to(tdt).contiguous()
        ragged = seg_offsets is not None
        m = 2 * int(order)
        if ragged:   # concatenated trajectories: waypoints [sum(S_b)+B,3], times [sum S_b], offsets [B+1] on the device
            if segment_major:
                raise ValueError("segment_major needs a uniform batch")
            self.off = seg_offsets.to(device=self.dev, dtype=torch.int64).contiguous()
            B, S = self.off.numel() - 1, 0
            total = self.tm.numel()
            if max_segments is None:
                max_segments = int((self.off[1:] - self.off[:-1]).max().item()) if B else 1
        else:
            B, S = self.tm.shape
        self.bc = (torch.zeros((1, 4, 3), dtype=tdt, device=self.dev) if bc is None
                   else bc.to(tdt).contiguous().reshape(-1, 4, 3))
        shape = (total, 3, m) if ragged else ((S, B, 3, m) if segment_major else (B, S, 3, m))
        self.out = out if out is not None else torch.empty(shape, dtype=tdt, device=self.dev)
        flags = ((FLAG_FORCE_GENERIC if force_generic else 0) | (FLAG_SEGMENT_MAJOR if segment_major else 0)
                 | (FLAG_NO_PERSISTENT if no_persistent else 0) | (FLAG_SPAN if span else 0))
        self.desc = make_desc(order, B, S, dtype, path_weight, vel_zero_weight, MEM_DEVICE, self.bc.shape[0] == B and B != 1,
                              seg_offsets_ptr=self.off.data_ptr() if ragged else None, max_segments=(max_segments or 0) if ragged else 0,
                              device_id=self.dev.index if self.dev.index is not None else -1, flags=flags)
        self.ws_bytes = workspace_bytes(self.desc)
        self.ws = torch.empty(max(self.ws_bytes, 1), dtype=torch.uint8, device=self.dev)
        self.kernel = kernel_name(self.desc)
        self._stream = stream
        self._args = (ctypes.byref(self.desc), self.wp.data_ptr(), self.tm.data_ptr(), self.bc.data_ptr(),
                      self.out.data_ptr(), None, None, self.ws.data_ptr() if self.ws_bytes else None, self.ws_bytes)

    def run(self, stream=None):
        import torch
        st = stream if stream is not None else (self._stream if self._stream is not None
                                                else torch.cuda.current_stream(self.dev).cuda_stream)
        rc = _lib.csp_minsnap_solve_batch(*self._args, ctypes.c_void_p(st))
        if rc:
            _check(rc)
        return self.out


def time_alloc_batch(waypoints, v_avg, min_time_s, seg_offsets=None, stream=None):
    """Batched T_i = max(|dp_i|/V_avg, min_time_s) (math_util/minimum_snap.cpp:63-72)."""
    ragged = seg_offsets is not None
    if _is_torch(waypoints):
        import torch
        dev, tdt = waypoints.device, waypoints.dtype
        dtype = DTYPE_F32 if tdt == torch.float32 else DTYPE_F64
        waypoints = waypoints.contiguous()
        if ragged:
            seg_offsets = seg_offsets.to(device=dev, dtype=torch.int64).contiguous()
            B = seg_offsets.numel() - 1
            times = torch.empty(waypoints.shape[0] - B, dtype=tdt, device=dev)
            S = 0
        else:
            B, S = waypoints.shape[0], waypoints.shape[1] - 1
            times = torch.empty((B, S), dtype=tdt, device=dev)
        desc = make_desc(1, B, S, dtype, mem_space=MEM_DEVICE,
                         seg_offsets_ptr=seg_offsets.data_ptr() if ragged else None, max_segments=1 if ragged else 0,
                         device_id=dev.index if dev.index is not None else -1)
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        _check(_lib.csp_minsnap_time_alloc_batch(ctypes.byref(desc), waypoints.data_ptr(), float(v_avg),
                                                 float(min_time_s), times.data_ptr(), ctypes.c_void_p(st)))
        return times
    waypoints = np.asarray(waypoints)
    dtype = DTYPE_F32 if waypoints.dtype == np.float32 else DTYPE_F64
    npdt = _np_dtype(dtype)
    waypoints = np.ascontiguousarray(waypoints, dtype=npdt)
    if ragged:
        seg_offsets = np.ascontiguousarray(seg_offsets, dtype=np.int64)
        B = seg_offsets.shape[0] - 1
        times = np.empty(waypoints.shape[0] - B, dtype=npdt)
        S = 0
    else:
        B, S = waypoints.shape[0], waypoints.shape[1] - 1
        times = np.empty((B, S), dtype=npdt)
    desc = make_desc(1, B, S, dtype, mem_space=MEM_HOST,
                     seg_offsets_ptr=seg_offsets.ctypes.data if ragged else None, max_segments=1 if ragged else 0)
    _check(_lib.csp_minsnap_time_alloc_batch(ctypes.byref(desc), waypoints.ctypes.data, float(v_avg),
                                             float(min_time_s), times.ctypes.data, None))
    return times


class Plan:
    __slots__ = ("times", "coeffs", "max_dev", "vel_zero_weight", "iterations", "status")


def plan_batch(waypoints, v_avg, min_time_s, bc=None, order=3, path_weight=0.0, vel_zero_weight=0.0):
    """Batched solver half of GenerateTrajectoryMatrix (math_util/minimum_snap.cpp:59-90): time
    allocation + the <=10x vel_zero_weight doubling loop.  Uniform batches, numpy (host) or torch
    CUDA tensors.  waypoints [B,S+1,3]."""
    on_device = _is_torch(waypoints)
    m = 2 * int(order)
    r = Plan()
    if on_device:
        import torch
        dev, tdt = waypoints.device, waypoints.dtype
        dtype = DTYPE_F32 if tdt == torch.float32 else DTYPE_F64
        waypoints = waypoints.contiguous()
        B, S = waypoints.shape[0], waypoints.shape[1] - 1
        bc = torch.zeros((1, 4, 3), dtype=tdt, device=dev) if bc is None else bc.to(tdt).contiguous().reshape(-1, 4, 3)
        r.times = torch.empty((B, S), dtype=tdt, device=dev)
        r.coeffs = torch.empty((B, S, 3, m), dtype=tdt, device=dev)
        r.max_dev = torch.empty(B, dtype=torch.float64, device=dev)
        r.vel_zero_weight = torch.empty(B, dtype=torch.float64, device=dev)
        r.iterations = torch.empty(B, dtype=torch.int32, device=dev)
        r.status = torch.empty(B, dtype=torch.int32, device=dev)
        desc = make_desc(order, B, S, dtype, path_weight, vel_zero_weight, MEM_DEVICE, bc.shape[0] == B and B != 1,
                         device_id=dev.index if dev.index is not None else -1)
        need = int(_lib.csp_minsnap_plan_workspace_bytes(ctypes.byref(desc)))
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        _check(_lib.csp_minsnap_plan_batch(ctypes.byref(desc), waypoints.data_ptr(), float(v_avg), float(min_time_s),
                                           bc.data_ptr(), r.times.data_ptr(), r.coeffs.data_ptr(), r.max_dev.data_ptr(),
                                           r.vel_zero_weight.data_ptr(), r.iterations.data_ptr(), r.status.data_ptr(),
                                           ws.data_ptr(), need, ctypes.c_void_p(st)))
        return r
    waypoints = np.asarray(waypoints)
    dtype = DTYPE_F32 if waypoints.dtype == np.float32 else DTYPE_F64
    npdt = _np_dtype(dtype)
    waypoints = np.ascontiguousarray(waypoints, dtype=npdt)
    B, S = waypoints.shape[0], waypoints.shape[1] - 1
    bc = np.zeros((1, 4, 3), dtype=npdt) if bc is None else np.ascontiguousarray(bc, dtype=npdt).reshape(-1, 4, 3)
    r.times = np.empty((B, S), dtype=npdt)
    r.coeffs = np.empty((B, S, 3, m), dtype=npdt)
    r.max_dev = np.empty(B, dtype=np.float64)
    r.vel_zero_weight = np.empty(B, dtype=np.float64)
    r.iterations = np.empty(B, dtype=np.int32)
    r.status = np.empty(B, dtype=np.int32)
    desc = make_desc(order, B, S, dtype, path_weight, vel_zero_weight, MEM_HOST, bc.shape[0] == B and B != 1)
    _check(_lib.csp_minsnap_plan_batch(ctypes.byref(desc), waypoints.ctypes.data, float(v_avg), float(min_time_s),
                                       bc.ctypes.data, r.times.ctypes.data, r.coeffs.ctypes.data, r.max_dev.ctypes.data,
                                       r.vel_zero_weight.ctypes.data, r.iterations.ctypes.data, r.status.ctypes.data,
                                       None, 0, None))
    return r


def sample_capacity(waypoints, v_avg, min_time_s, order=3):
    """Upper bound of the samples per trajectory (csp_minsnap_sample_capacity), from host waypoints [B,S+1,3]."""
    waypoints = np.asarray(waypoints)
    dtype = DTYPE_F32 if waypoints.dtype == np.float32 else DTYPE_F64
    waypoints = np.ascontiguousarray(waypoints, dtype=_np_dtype(dtype))
    desc = make_desc(order, waypoints.shape[0], waypoints.shape[1] - 1, dtype, mem_space=MEM_HOST)
    cap = int(_lib.csp_minsnap_sample_capacity(ctypes.byref(desc), waypoints.ctypes.data, float(v_avg), float(min_time_s)))
    if cap < 0:
        raise CspError(-1, "csp_minsnap_sample_capacity")
    return cap


class Generated(Plan):
    """Plan + samples [B,capacity,3], counts [B], stats [B,2]."""
    __slots__ = ("samples", "counts", "stats")


def generate_batch(waypoints, v_avg, min_time_s, sample_distance, capacity=None, bc=None, order=3, path_weight=0.0,
                   vel_zero_weight=0.0, long_segments=False):
    """The whole of GenerateTrajectoryMatrix (math_util/minimum_snap.cpp:22-206) in one call
    (csp_minsnap_generate_batch = plan_batch + sample_batch, bit for bit).  Uniform batches, numpy (host: one upload,
    one download, one synchronisation) or torch CUDA tensors (asynchronous; `capacity` required)."""
    on_device = _is_torch(waypoints)
    m = 2 * int(order)
    r = Generated()
    flags = FLAG_LONG_SEGMENTS if long_segments else 0
    if on_device:
        import torch
        if capacity is None:
            raise ValueError("device-memory generate_batch needs a capacity")
        dev, tdt = waypoints.device, waypoints.dtype
        dtype = DTYPE_F32 if tdt == torch.float32 else DTYPE_F64
        waypoints = waypoints.contiguous()
        B, S = waypoints.shape[0], waypoints.shape[1] - 1
        bc = torch.zeros((1, 4, 3), dtype=tdt, device=dev) if bc is None else bc.to(tdt).contiguous().reshape(-1, 4, 3)
        r.times = torch.empty((B, S), dtype=tdt, device=dev)
        r.coeffs = torch.empty((B, S, 3, m), dtype=tdt, device=dev)
        r.max_dev = torch.empty(B, dtype=torch.float64, device=dev)
        r.vel_zero_weight = torch.empty(B, dtype=torch.float64, device=dev)
        r.iterations = torch.empty(B, dtype=torch.int32, device=dev)
        r.status = torch.empty(B, dtype=torch.int32, device=dev)
        r.samples = torch.zeros((B, capacity, 3), dtype=tdt, device=dev)
        r.counts = torch.empty(B, dtype=torch.int32, device=dev)
        r.stats = torch.empty((B, 2), dtype=torch.float64, device=dev)
        desc = make_desc(order, B, S, dtype, path_weight, vel_zero_weight, MEM_DEVICE, bc.shape[0] == B and B != 1,
                         device_id=dev.index if dev.index is not None else -1, flags=flags)
        need = int(_lib.csp_minsnap_plan_workspace_bytes(ctypes.byref(desc)))
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        _check(_lib.csp_minsnap_generate_batch(ctypes.byref(desc), waypoints.data_ptr(), float(v_avg), float(min_time_s), bc.data_ptr(),
                                               float(sample_distance), int(capacity), r.samples.data_ptr(), r.counts.data_ptr(),
                                               r.stats.data_ptr(), r.times.data_ptr(), r.coeffs.data_ptr(), r.max_dev.data_ptr(),
                                               r.vel_zero_weight.data_ptr(), r.iterations.data_ptr(), r.status.data_ptr(),
                                               ws.data_ptr(), need, ctypes.c_void_p(st)))
        return r
    waypoints = np.asarray(waypoints)
    dtype = DTYPE_F32 if waypoints.dtype == np.float32 else DTYPE_F64
    npdt = _np_dtype(dtype)
    waypoints = np.ascontiguousarray(waypoints, dtype=npdt)
    B, S = waypoints.shape[0], waypoints.shape[1] - 1
    bc = np.zeros((1, 4, 3), dtype=npdt) if bc is None else np.ascontiguousarray(bc, dtype=npdt).reshape(-1, 4, 3)
    desc = make_desc(order, B, S, dtype, path_weight, vel_zero_weight, MEM_HOST, bc.shape[0] == B and B != 1, flags=flags)
    if capacity is None:
        capacity = int(_lib.csp_minsnap_sample_capacity(ctypes.byref(desc), waypoints.ctypes.data, float(v_avg), float(min_time_s)))
    r.times = np.empty((B, S), dtype=npdt)
    r.coeffs = np.empty((B, S, 3, m), dtype=npdt)
    r.max_dev = np.empty(B, dtype=np.float64)
    r.vel_zero_weight = np.empty(B, dtype=np.float64)
    r.iterations = np.empty(B, dtype=np.int32)
    r.status = np.empty(B, dtype=np.int32)
    r.samples = np.zeros((B, capacity, 3), dtype=npdt)
    r.counts = np.empty(B, dtype=np.int32)
    r.stats = np.empty((B, 2), dtype=np.float64)
    _check(_lib.csp_minsnap_generate_batch(ctypes.byref(desc), waypoints.ctypes.data, float(v_avg), float(min_time_s), bc.ctypes.data,
                                           float(sample_distance), int(capacity), r.samples.ctypes.data, r.counts.ctypes.data,
                                           r.stats.ctypes.data, r.times.ctypes.data, r.coeffs.ctypes.data, r.max_dev.ctypes.data,
                                           r.vel_zero_weight.ctypes.data, r.iterations.ctypes.data, r.status.ctypes.data,
                                           None, 0, None))
    return r


def sample_batch(times, coeffs, sample_distance, capacity, order=None, out=None, one_lane=False, long_segments=False,
                 seg_offsets=None):
    """Batched sampling half of GenerateTrajectoryMatrix (math_util/minimum_snap.cpp:97-205).
    times [B,S], coeffs [B,S,3,2o].  Returns (samples [B,capacity,3], counts [B], stats [B,2]).
    `out` (device path): a (samples, counts, stats) triple to reuse; rows beyond counts[b] are then
    left as they were instead of zero.  `one_lane` forces the one-lane-per-trajectory kernel (A/B tests);
    `long_segments` (device path) selects the wave-per-trajectory kernel for legs of hundreds of candidates
    (the host path decides from the times).  Ragged batches (host arrays): times [sum S_b], coeffs [sum S_b,3,2o],
    `seg_offsets` [B+1]."""
    on_device = _is_torch(times)
    if seg_offsets is not None:
        if on_device:
            raise ValueError("ragged sampling takes host arrays here")
        seg_offsets = np.ascontiguousarray(seg_offsets, dtype=np.int64)
        B, S = seg_offsets.shape[0] - 1, 0
    else:
        B, S = times.shape
    order = int(order) if order is not None else int(coeffs.shape[-1]) // 2
    if on_device:
        import torch
        dev, tdt = times.device, times.dtype
        dtype = DTYPE_F32 if tdt == torch.float32 else DTYPE_F64
        times, coeffs = times.contiguous(), coeffs.to(tdt).contiguous()
        if out is not None:
            samples, counts, stats = out
        else:
            samples = torch.zeros((B, capacity, 3), dtype=tdt, device=dev)
            counts = torch.empty(B, dtype=torch.int32, device=dev)
            stats = torch.empty((B, 2), dtype=torch.float64, device=dev)
        desc = make_desc(order, B, S, dtype, mem_space=MEM_DEVICE, device_id=dev.index if dev.index is not None else -1,
                         flags=(FLAG_FORCE_GENERIC if one_lane else 0) | (FLAG_LONG_SEGMENTS if long_segments else 0))
        st = torch.cuda.current_stream(dev).cuda_stream
        _check(_lib.csp_minsnap_sample_batch(ctypes.byref(desc), times.data_ptr(), coeffs.data_ptr(), float(sample_distance),
                                             int(capacity), samples.data_ptr(), counts.data_ptr(), stats.data_ptr(),
                                             ctypes.c_void_p(st)))
        return samples, counts, stats
    times = np.asarray(times)
    dtype = DTYPE_F32 if times.dtype == np.float32 else DTYPE_F64
    npdt = _np_dtype(dtype)
    times = np.ascontiguousarray(times, dtype=npdt)
    coeffs = np.ascontiguousarray(coeffs, dtype=npdt)
    samples = np.zeros((B, capacity, 3), dtype=npdt)
    counts = np.empty(B, dtype=np.int32)
    stats = np.empty((B, 2), dtype=np.float64)
    desc = make_desc(order, B, S, dtype, mem_space=MEM_HOST,
                     seg_offsets_ptr=seg_offsets.ctypes.data if seg_offsets is not None else None,
                     max_segments=int(np.max(np.diff(seg_offsets))) if seg_offsets is not None and B else 0,
                     flags=(FLAG_FORCE_GENERIC if one_lane else 0) | (FLAG_LONG_SEGMENTS if long_segments else 0))
    _check(_lib.csp_minsnap_sample_batch(ctypes.byref(desc), times.ctypes.data, coeffs.ctypes.data, float(sample_distance),
                                         int(capacity), samples.ctypes.data, counts.ctypes.data, stats.ctypes.data, None))
    return samples, counts, stats


def _geo(fn, pts, ref):
    ref = np.ascontiguousarray(ref, dtype=np.float64).reshape(3)
    if _is_torch(pts):
        import torch
        pts = pts.to(torch.float64).contiguous()
        out = torch.empty_like(pts)
        st = torch.cuda.current_stream(pts.device).cuda_stream
        _check(fn(pts.data_ptr(), ref.ctypes.data, out.data_ptr(), pts.shape[0], MEM_DEVICE,
                  pts.device.index if pts.device.index is not None else -1, ctypes.c_void_p(st)))
        return out
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
    out = np.empty_like(pts)
    _check(fn(pts.ctypes.data, ref.ctypes.data, out.ctypes.data, pts.shape[0], MEM_HOST, -1, None))
    return out


def wgs84_to_enu_batch(lla, ref):
    """Batched UavPathPlanner::wgs84ToENU (uavPathPlanning.cpp:1046-1063, :1085-1095).
    lla [N,3] = (lon_deg, lat_deg, alt_m); ref [3] same convention."""
    return _geo(_lib.csp_geo_wgs84_to_enu_batch, lla, ref)


def enu_to_wgs84_batch(enu, ref):
    """Batched UavPathPlanner::enuToWGS84 (uavPathPlanning.cpp:1066-1083, :1098-1108)."""
    return _geo(_lib.csp_geo_enu_to_wgs84_batch, enu, ref)


class AltParams(ctypes.Structure):
    """Mirror of `csp_alt_params` (include/csp_alt.h; reference AltitudeParams, uavPathPlanning.hpp:415-421)."""
    _fields_ = [("lambda_smooth", ctypes.c_double), ("lambda_follow", ctypes.c_double),
                ("safe_distance", ctypes.c_double), ("max_climb_rate", ctypes.c_double)]


_lib.csp_alt_workspace_bytes.restype = ctypes.c_size_t
_lib.csp_alt_workspace_bytes.argtypes = [ctypes.c_int64]
_lib.csp_alt_optimize_heights_batch.restype = ctypes.c_int
_lib.csp_alt_optimize_heights_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                ctypes.POINTER(AltParams), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                                ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p]
_lib.csp_alt_global_smooth_batch.restype = ctypes.c_int
_lib.csp_alt_global_smooth_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                             ctypes.POINTER(AltParams), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p]


def _alt_params(lambda_smooth, lambda_follow, safe_distance, max_climb_rate):
    return AltParams(float(lambda_smooth), float(lambda_follow), float(safe_distance), float(max_climb_rate))


def _alt_ws_bytes(total):
    _lib.csp_alt_workspace_bytes.restype = ctypes.c_size_t
    _lib.csp_alt_workspace_bytes.argtypes = [ctypes.c_int64]
    return int(_lib.csp_alt_workspace_bytes(int(total)))


def alt_optimize_heights_batch(xyz, elev, offsets, lambda_smooth=1.0, lambda_follow=0.0, safe_distance=50.0,
                               max_climb_rate=2.0):
    """Batched UavPathPlanner::optimizeHeights (uavPathPlanning.cpp:1575-1713).  numpy arrays (host memory) or torch CUDA
    tensors (device memory): xyz [total,3], elev [total] (NaN = no terrain sample), offsets [B+1].  Returns z [total]."""
    p = _alt_params(lambda_smooth, lambda_follow, safe_distance, max_climb_rate)
    if _is_torch(xyz):
        import torch
        dev = xyz.device
        xyz = xyz.to(torch.float64).contiguous().reshape(-1, 3)
        elev = elev.to(torch.float64).contiguous()
        offsets = offsets.to(device=dev, dtype=torch.int64).contiguous()
        out = torch.empty(xyz.shape[0], dtype=torch.float64, device=dev)
        need = _alt_ws_bytes(xyz.shape[0])
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
        _check(_lib.csp_alt_optimize_heights_batch(xyz.data_ptr(), elev.data_ptr(), offsets.data_ptr(), offsets.numel() - 1, ctypes.byref(p),
                                                   out.data_ptr(), ws.data_ptr(), need, MEM_DEVICE, dev.index if dev.index is not None else -1,
                                                   ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        torch.cuda.current_stream(dev).synchronize()   # `ws` must outlive the kernel
        return out
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    elev = np.ascontiguousarray(elev, dtype=np.float64)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    out = np.empty(xyz.shape[0])
    _check(_lib.csp_alt_optimize_heights_batch(xyz.ctypes.data, elev.ctypes.data, offsets.ctypes.data, offsets.shape[0] - 1,
                                               ctypes.byref(p), out.ctypes.data, None, 0, MEM_HOST, -1, None))
    return out


def alt_global_smooth_batch(input_z, xyz, offsets, lambda_smooth=1.0, max_climb_rate=2.0):
    """Batched UavPathPlanner::optimizeHeightsGlobalSmooth (uavPathPlanning.cpp:1715-1827).
    Returns (z [total], solves [B]); numpy (host memory) or torch CUDA tensors (device memory)."""
    p = _alt_params(lambda_smooth, 0.0, 0.0, max_climb_rate)
    if _is_torch(xyz):
        import torch
        dev = xyz.device
        xyz = xyz.to(torch.float64).contiguous().reshape(-1, 3)
        input_z = input_z.to(torch.float64).contiguous()
        offsets = offsets.to(device=dev, dtype=torch.int64).contiguous()
        out = torch.empty(xyz.shape[0], dtype=torch.float64, device=dev)
        solves = torch.empty(offsets.numel() - 1, dtype=torch.int32, device=dev)
        need = _alt_ws_bytes(xyz.shape[0])
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
        _check(_lib.csp_alt_global_smooth_batch(input_z.data_ptr(), xyz.data_ptr(), offsets.data_ptr(), offsets.numel() - 1, ctypes.byref(p),
                                                out.data_ptr(), solves.data_ptr(), ws.data_ptr(), need, MEM_DEVICE,
                                                dev.index if dev.index is not None else -1,
                                                ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        torch.cuda.current_stream(dev).synchronize()
        return out, solves
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    input_z = np.ascontiguousarray(input_z, dtype=np.float64)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    out = np.empty(xyz.shape[0])
    solves = np.empty(offsets.shape[0] - 1, dtype=np.int32)
    _check(_lib.csp_alt_global_smooth_batch(input_z.ctypes.data, xyz.ctypes.data, offsets.ctypes.data, offsets.shape[0] - 1,
                                            ctypes.byref(p), out.ctypes.data, solves.ctypes.data, None, 0, MEM_HOST, -1, None))
    return out, solves


_lib.csp_bezier_generate_batch.restype = ctypes.c_int
_lib.csp_bezier_generate_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32,
                                           ctypes.c_void_p]


def bezier_generate_batch(waypoints, offsets, resolution=1.0, min_radius=1.0, capacity=4096):
    """Batched math_util::Bezier::GenerateTrajectoryMatrix (math_util/bezier.cpp:127-190; include/csp_bezier.h).
    waypoints [total,3] (paths concatenated), offsets [B+1] point prefix sums; numpy (host) or torch CUDA tensors.
    Returns (samples [B,capacity,3], counts [B])."""
    if _is_torch(waypoints):
        import torch
        dev = waypoints.device
        wp = waypoints.to(torch.float64).contiguous()
        off = offsets.to(device=dev, dtype=torch.int64).contiguous()
        B = off.numel() - 1
        samples = torch.zeros((B, capacity, 3), dtype=torch.float64, device=dev)
        counts = torch.empty(B, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        _check(_lib.csp_bezier_generate_batch(wp.data_ptr(), off.data_ptr(), B, float(resolution), float(min_radius), int(capacity),
                                              samples.data_ptr(), counts.data_ptr(), MEM_DEVICE,
                                              dev.index if dev.index is not None else -1, ctypes.c_void_p(st)))
        return samples, counts
    wp = np.ascontiguousarray(waypoints, dtype=np.float64).reshape(-1, 3)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    B = off.shape[0] - 1
    samples = np.zeros((B, capacity, 3))
    counts = np.empty(B, dtype=np.int32)
    _check(_lib.csp_bezier_generate_batch(wp.ctypes.data, off.ctypes.data, B, float(resolution), float(min_radius), int(capacity),
                                          samples.ctypes.data, counts.ctypes.data, MEM_HOST, -1, None))
    return samples, counts
