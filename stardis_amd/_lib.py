"""ctypes binding of libstardis_hip.so (include/stardis_hip.h) and the device-array helper.

There is no CPU fallback anywhere in this package: if the HIP library is missing or no
MI355X is visible, every compute entry point raises.
"""
import ctypes as C
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# STARDIS_AMD_LIB points at another build of the same ABI (A/B measurements of kernel changes)
LIB_PATH = os.environ.get("STARDIS_AMD_LIB") or os.path.join(_HERE, "lib", "libstardis_hip.so")

c_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_i64 = C.c_int64
_int = C.c_int


class Continuum(C.Structure):
    """struct sdx_continuum (include/stardis_hip.h)"""

    _fields_ = [
        ("lambdas", _vp),
        ("n_table", _int),
        ("table_wavelength", _vp),
        ("table_sigma", _vp),
        ("table_density", _vp),
        ("bf_n_species", _int),
        ("bf_n_levels", _int),
        ("bf_species_offsets", _vp),
        ("bf_species_ion_number", _vp),
        ("bf_cutoff", _vp),
        ("bf_level_density", _vp),
        ("ff_n_species", _int),
        ("ff_species_ion_number", _vp),
        ("ff_number_density", _vp),
        ("ray_n_h", _vp),
        ("ray_n_he", _vp),
        ("ray_n_h2", _vp),
        ("rayleigh_enabled", _int),
        ("electron_density", _vp),
        ("temperature", _vp),
        ("n_file_planes", _int),
        ("file_plane", _vp * 4),
        ("file_plane_ld", _i64),
    ]


class LineListStruct(C.Structure):
    """struct sdx_linelist (include/stardis_hip.h)"""

    _fields_ = [
        ("n_lines", _i64),
        ("nu", _vp),
        ("e_low_ev", _vp),
        ("g_lo", _vp),
        ("strength", _vp),
        ("pop_row", _vp),
        ("pop", _vp),
        ("n_pop_rows", _int),
        ("alpha_coefficient", C.c_double),
        ("mass", _vp),
        ("microturbulence", C.c_double),
        ("gamma_mode", _int),
        ("broadening_flags", _int),
        ("atomic_number", _vp),
        ("ion_number", _vp),
        ("ionization_energy", _vp),
        ("upper_energy", _vp),
        ("lower_energy", _vp),
        ("A_ul", _vp),
        ("stark", _vp),
        ("waals", _vp),
        ("temperature", _vp),
        ("electron_density", _vp),
        ("h_density", _vp),
    ]


class SynthesisOptions(C.Structure):
    """struct sdx_synthesis_options (include/stardis_hip.h)"""

    _fields_ = [
        ("source", _vp),
        ("source_ld", _i64),
        ("I_nus", _vp),
        ("inward_rays", _int),
        ("photospheric_correction", C.c_double),
        ("n_line_planes", _int),
        ("line_plane", _vp * 2),
        ("line_plane_ld", _i64),
        ("linelist", C.POINTER(LineListStruct)),
        ("line_m_max", _vp),
    ]


# name -> (restype, argtypes); every function declared in include/stardis_hip.h
PROTOTYPES = {
    "sdx_version": (C.c_char_p, []),
    "sdx_last_error_string": (C.c_char_p, []),
    "sdx_last_error_code": (_int, []),
    "sdx_group_create": (_vp, [_int, C.POINTER(_int)]),
    "sdx_group_destroy": (None, [_vp]),
    "sdx_group_size": (_int, [_vp]),
    "sdx_group_context": (_vp, [_vp, _int]),
    "sdx_group_last_gather": (_int, [_vp, C.POINTER(_int), C.POINTER(_i64), C.POINTER(_int)]),
    "sdx_synthesize_sharded_f64": (_int, [_vp, _int, _i64, _vp, _i64, _vp, _vp, _vp, _int, _vp, C.POINTER(Continuum), _int, _vp, _vp, _vp, _vp,
                                          _vp, _vp, _vp, _vp, _vp]),
    "sdx_device_count": (_int, []),
    "sdx_set_device": (_int, [_int]),
    "sdx_create": (_vp, [_int, _vp]),
    "sdx_destroy": (None, [_vp]),
    "sdx_set_stream": (_int, [_vp, _vp]),
    "sdx_get_stream": (_vp, [_vp]),
    "sdx_synchronize": (_int, [_vp]),
    "sdx_set_int_option": (_int, [_vp, C.c_char_p, _i64]),
    "sdx_far_field_active": (_int, [_vp, _i64]),
    "sdx_far_field_rule": (_int, [C.POINTER(_i64), C.POINTER(_int), C.POINTER(_int)]),
    "sdx_malloc": (_vp, [_vp, C.c_size_t]),
    "sdx_free": (_int, [_vp, _vp]),
    "sdx_memcpy_h2d": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "sdx_memcpy_d2h": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "sdx_memset": (_int, [_vp, _vp, _int, C.c_size_t]),
    "sdx_host_alloc": (_vp, [_vp, C.c_size_t]),
    "sdx_host_free": (_int, [_vp]),
    "sdx_memcpy_h2d_pinned": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "sdx_memcpy_d2h_pinned": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "sdx_reserve_line_workspace": (_int, [_vp, _int, _i64]),
    "sdx_graph_begin": (_int, [_vp]),
    "sdx_graph_end": (_int, [_vp, C.POINTER(_vp)]),
    "sdx_graph_launch": (_int, [_vp, _vp]),
    "sdx_graph_destroy": (_int, [_vp, _vp]),
    "sdx_timer_start": (_int, [_vp]),
    "sdx_timer_stop": (_int, [_vp, c_dp]),
    "sdx_profile_enable": (_int, [_vp, _int]),
    "sdx_profile_reset": (_int, [_vp]),
    "sdx_profile_variant": (_int, [_vp, C.c_char_p, C.c_char_p, _int]),
    "sdx_profile_get": (_int, [_vp, C.c_char_p, C.POINTER(_i64), c_dp]),
    "sdx_line_opacity_dev": (_int, [_vp, _int, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _int, _vp, _vp, _i64, _int, _vp]),
    "sdx_line_opacity_f64": (_int, [_vp, _int, _i64, _vp, _i64, _vp, _vp, _vp, _int, _vp, _vp, C.POINTER(_i64)]),
    "sdx_line_opacity_f32mix": (_int, [_vp, _int, _i64, _vp, _i64, _vp, _vp, _vp, _int, _vp, _vp, C.POINTER(_i64)]),
    "sdx_raytrace_f32mix": (_int, [_vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdx_synthesize_f32mix": (_int, [_vp, _int, _i64, _vp, _i64, _vp, _vp, _vp, _int, _vp, C.POINTER(Continuum), _int, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _vp]),
    "sdx_line_windows_dev": (_int, [_vp, _int, _i64, _vp, _i64, _vp, _vp, _vp, _int, _vp, _vp, _vp]),
    "sdx_faddeeva_dev": (_int, [_vp, _i64, _vp, _vp]),
    "sdx_voigt_profile_dev": (_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "sdx_voigt_term_dev": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "sdx_voigt_term_f32_dev": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "sdx_calc_gamma_dev": (_int, [_vp, _i64, _int] + [_vp] * 9 + [_int, _vp]),
    "sdx_doppler_widths_dev": (_int, [_vp, _i64, _int, _vp, _vp, _vp, C.c_double, _vp]),
    "sdx_calc_vald_gamma_dev": (_int, [_vp, _i64, _int] + [_vp] * 12 + [_int, _vp]),
    "sdx_broadening_scalar_dev": (_int, [_vp, _int, _i64] + [_vp] * 6),
    "sdx_alpha_file_1d_dev": (_int, [_vp, _int, _i64, _vp, _int, _vp, _vp, _vp, _vp, _i64]),
    "sdx_sigma_table_2d_dev": (_int, [_vp, _int, _vp, _int, _vp, _vp, _vp, _vp, _int, _i64, _vp, _vp, _int, _vp, _vp, _i64, _vp]),
    "sdx_alpha_file_2d_dev": (_int, [_vp, _int, _i64, _vp, _i64, _vp, _vp, _i64]),
    "sdx_alpha_bf_dev": (_int, [_vp, _int, _i64, _vp, _int, _vp, _vp, _vp, _vp, _vp, _i64]),
    "sdx_alpha_ff_dev": (_int, [_vp, _int, _i64, _vp, _vp, _int, _vp, _vp, _vp, _i64]),
    "sdx_alpha_rayleigh_dev": (_int, [_vp, _int, _i64, _vp, _vp, _vp, _vp, _vp, _i64]),
    "sdx_alpha_electron_dev": (_int, [_vp, _int, _i64, _vp, _vp, _i64]),
    "sdx_accumulate_dev": (_int, [_vp, _int, _i64, _vp, _i64, _vp, _i64]),
    "sdx_blackbody_dev": (_int, [_vp, _int, _i64, _vp, _vp, _vp, _i64]),
    "sdx_calc_weights_dev": (_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "sdx_raytrace_dev": (_int, [_vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _int]),
    "sdx_raytrace_spherical_dev": (_int, [_vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _int, C.c_double]),
    "sdx_raytrace_source_dev": (_int, [_vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _int, _int, C.c_double]),
    "sdx_raytrace_f64": (_int, [_vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdx_total_alphas_dev": (_int, [_vp, _int, _i64, _vp, _i64, _i64, C.POINTER(Continuum), _vp, _i64, _vp, _i64]),
    "sdx_convolve1d_reflect_dev": (_int, [_vp, _i64, _vp, _int, _vp, _int, _vp]),
    "sdx_flux_nu_to_lambda_dev": (_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "sdx_synthesize_dev": (_int, [_vp, _int, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _int, _vp, C.POINTER(Continuum), _int,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sdx_synthesize_ex_dev": (_int, [_vp, _int, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _int, _vp, C.POINTER(Continuum), _int,
                                     _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "sdx_synthesize_opt_dev": (_int, [_vp, _int, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _int, _vp, C.POINTER(Continuum), _int,
                                      _vp, _vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(SynthesisOptions), _vp]),
    "sdx_synthesize_classify_dev": (_int, [_vp, _int, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _int, _vp, C.POINTER(Continuum), _i64, _i64, _vp]),
    "sdx_synthesize_f64": (_int, [_vp, _int, _i64, _vp, _i64, _vp, _vp, _vp, _int, _vp, C.POINTER(Continuum), _int, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _vp]),
    "sdx_continuum_f64": (_int, [_vp, _int, _i64, _vp, C.POINTER(Continuum), _vp, _vp, _vp, _vp, _vp, _vp]),
    "sdx_alpha_line_levels_dev": (_int, [_vp, _i64, _int, _int, _vp, _vp, _vp, _vp, C.c_double, _vp]),
    "sdx_line_params_dev": (_int, [_vp, _int, C.POINTER(LineListStruct), _vp, _vp, _vp]),
    "sdx_line_opacity_linelist_dev": (_int, [_vp, _int, _i64, _vp, _i64, _i64, C.POINTER(LineListStruct), _vp, _i64, _int, _vp]),
    "sdx_synthesize_linelist_dev": (_int, [_vp, _int, _i64, _vp, _i64, _i64, C.POINTER(LineListStruct), C.POINTER(Continuum), _int,
                                           _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
}

_lib = None
_lock = threading.RLock()
_contexts = {}


def load():
    """The shared library, or RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} is missing: build it with `make -C stardis_amd/csrc` "
                        "(or __graft_entry__.build()); stardis_amd has no CPU fallback"
                    )
                lib = C.CDLL(LIB_PATH)
                for name, (res, args) in PROTOTYPES.items():
                    fn = getattr(lib, name)
                    fn.restype = res
                    fn.argtypes = args
                _lib = lib
    return _lib


class StaleGraphError(RuntimeError):
    """sdx_graph_launch refused a graph captured before the context's scratch was reallocated (SDX_ERR_STALE)."""


class CommError(RuntimeError):
    """An RCCL failure inside the library (SDX_ERR_COMM): library not loadable, communicator set-up, the collective."""


def check(rc):
    if rc == 0:
        return
    msg = load().sdx_last_error_string().decode()
    if rc == -1:
        raise ValueError(msg)
    if rc == -4:
        raise MemoryError(msg)
    if rc == -3:
        raise CommError(msg)
    if rc == -5:
        raise StaleGraphError(msg)
    raise RuntimeError(f"stardis_hip error {rc}: {msg}")


class Context:
    """One sdx_ctx: a device, a stream and the library's scratch."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        if self.lib.sdx_device_count() <= device:
            raise RuntimeError(
                f"no HIP device {device} visible (sdx_device_count() = {self.lib.sdx_device_count()}); "
                "the STARDIS hot path of stardis_amd runs on MI355X only"
            )
        self.device = device
        self.handle = self.lib.sdx_create(device, stream)
        if not self.handle:
            raise RuntimeError(self.lib.sdx_last_error_string().decode())

    def close(self):
        if getattr(self, "handle", None):
            if getattr(self, "_pinned", None) is not None:
                self._pinned.release_idle()
            self.lib.sdx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self.lib.sdx_synchronize(self.handle))

    def set_option(self, name, value):
        check(self.lib.sdx_set_int_option(self.handle, name.encode(), int(value)))

    # -- arrays -------------------------------------------------------------------------------
    def empty(self, shape, dtype=np.float64):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype=np.float64):
        a = DeviceArray(self, shape, dtype)
        check(self.lib.sdx_memset(self.handle, a.ptr, 0, a.nbytes))
        return a

    def upload(self, array, dtype=np.float64):
        host = np.ascontiguousarray(array, dtype=dtype)
        a = DeviceArray(self, host.shape, dtype)
        check(self.lib.sdx_memcpy_h2d(self.handle, a.ptr, host.ctypes.data, host.nbytes))
        return a

    def call(self, name, *args):
        check(getattr(self.lib, name)(self.handle, *args))

    def profile(self, kernel):
        """(launches, milliseconds) recorded under `kernel` since sdx_profile_reset (sdx_profile_enable(1) switches recording on)."""
        cnt, ms = C.c_int64(), C.c_double()
        check(self.lib.sdx_profile_get(self.handle, kernel.encode(), C.byref(cnt), C.byref(ms)))
        return cnt.value, ms.value

    def profile_variant(self, kernel):
        """which device kernel / configuration ran under a stage name in the profiled launches ("" when the stage has one form only)"""
        buf = C.create_string_buffer(96)
        check(self.lib.sdx_profile_variant(self.handle, kernel.encode(), buf, 96))
        return buf.value.decode()

    @property
    def pinned(self):
        """The context's pool of page-locked host arrays (PinnedPool)."""
        pool = getattr(self, "_pinned", None)
        if pool is None:
            pool = self._pinned = PinnedPool(self)
        return pool


class PinnedPool:
    """numpy arrays in page-locked host memory (sdx_host_alloc), for buffers the device reads or writes by DMA: a staging area
    that lives as long as the pool, and result arrays whose memory comes back to the pool when the last view of them is gone
    (idle blocks are kept up to `keep` bytes: hipHostMalloc costs ~0.1 ms).  No more than `limit` bytes are out at a time;
    beyond that `empty` returns None and the caller uses pageable memory and the bounce-buffer copies."""

    def __init__(self, ctx, keep=256 << 20, limit=1 << 30):
        self.ctx, self.keep, self.limit = ctx, keep, limit  # bytes of idle blocks kept; bytes handed out at most
        self._free = {}
        self._idle = 0
        self._out = 0
        self._lock = threading.Lock()

    BIGGEST = 128 << 20  # larger arrays are not worth page-locking per call: they take the pageable path

    @staticmethod
    def _capacity(nbytes):
        """Size class: powers of two up to 1 MiB, whole MiB above."""
        nbytes = int(nbytes)
        return max(4096, 1 << (nbytes - 1).bit_length()) if nbytes <= (1 << 20) else (nbytes + (1 << 20) - 1) & ~((1 << 20) - 1)

    def _take(self, cap):
        with self._lock:
            if self._out + cap > self.limit:
                return None
            blocks = self._free.get(cap)
            ptr = blocks.pop() if blocks else None
            if ptr is not None:
                self._idle -= cap
            self._out += cap
        if ptr is None:
            ptr = self.ctx.lib.sdx_host_alloc(self.ctx.handle, cap)
            if not ptr:
                with self._lock:
                    self._out -= cap
                return None
        return ptr

    def _give_back(self, ptr, cap):
        with self._lock:
            self._out -= cap
            if self._idle + cap <= self.keep:
                self._free.setdefault(cap, []).append(ptr)
                self._idle += cap
                return
        try:
            self.ctx.lib.sdx_host_free(ptr)
        except Exception:
            pass

    def release_idle(self):
        """Free the idle blocks (Context.close); blocks still held by arrays are freed when those arrays go."""
        with self._lock:
            blocks = [p for v in self._free.values() for p in v]
            self._free.clear()
            self._idle = 0
            self.keep = 0
        for p in blocks:
            try:
                self.ctx.lib.sdx_host_free(p)
            except Exception:
                pass

    def empty(self, shape, dtype=np.float64):
        dtype = np.dtype(dtype)
        count = int(np.prod(shape, dtype=np.int64))
        nbytes = max(count * dtype.itemsize, 1)
        if nbytes > self.BIGGEST:
            return None
        cap = self._capacity(nbytes)
        ptr = self._take(cap)
        if ptr is None:
            return None
        # (the ctypes array type is made per size CLASS — ctypes keeps every array type it ever built — and the numpy view is cut
        # from it: a caller that asks for many different sizes does not grow that cache without bound)
        block = (C.c_char * cap).from_address(ptr)
        weakref.finalize(block, self._give_back, ptr, cap)  # every view of the array keeps `block` alive through .base
        return np.frombuffer(block, dtype=dtype, count=count).reshape(shape)


class DeviceArray:
    """A device buffer allocated through the C ABI (sdx_malloc): pointer + shape, nothing more."""

    def __init__(self, ctx, shape, dtype=np.float64):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.ptr = ctx.lib.sdx_malloc(ctx.handle, max(self.nbytes, 8))
        if not self.ptr:
            raise MemoryError(ctx.lib.sdx_last_error_string().decode())

    def numpy(self):
        # planes land in page-locked memory by DMA (PinnedPool: back to the pool when the array is dropped); small arrays and
        # whatever exceeds the pool's limit take the bounce-buffer copy into pageable memory
        out = self.ctx.pinned.empty(self.shape, self.dtype) if self.nbytes >= (64 << 10) else None
        if out is not None:
            check(self.ctx.lib.sdx_memcpy_d2h_pinned(self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes))
            return out
        out = np.empty(self.shape, dtype=self.dtype)
        check(self.ctx.lib.sdx_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def set(self, array):
        host = np.ascontiguousarray(array, dtype=self.dtype)
        assert host.shape == self.shape
        check(self.ctx.lib.sdx_memcpy_h2d(self.ctx.handle, self.ptr, host.ctypes.data, host.nbytes))

    def zero(self):
        check(self.ctx.lib.sdx_memset(self.ctx.handle, self.ptr, 0, self.nbytes))

    def free(self):
        if self.ptr and self.ctx.handle:
            self.ctx.lib.sdx_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def default_context():
    """Process-wide context on the device chosen by LOCAL_RANK (one process per GPU) or device 0."""
    dev = int(os.environ.get("STARDIS_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    ctx = _contexts.get(dev)
    if ctx is None:
        with _lock:
            ctx = _contexts.get(dev)
            if ctx is None:
                ctx = Context(dev)
                _contexts[dev] = ctx
    return ctx


def ptr_of(x):
    """Device address of a DeviceArray, a torch CUDA tensor or None."""
    if x is None:
        return None
    if isinstance(x, DeviceArray):
        return x.ptr
    if hasattr(x, "data_ptr"):
        if not x.is_cuda or not x.is_contiguous():
            raise ValueError("torch tensors passed to stardis_amd must be contiguous CUDA tensors")
        return x.data_ptr()
    raise TypeError(f"not a device array: {type(x)!r}")


def plain(x):
    """Strip astropy units / pandas wrappers the way numba does at its boundary."""
    if hasattr(x, "unit") and hasattr(x, "value"):  # astropy Quantity
        x = x.value
    elif hasattr(x, "to_numpy"):  # pandas Series / DataFrame
        x = x.to_numpy()
    return np.asarray(x)
