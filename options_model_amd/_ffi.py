"""ctypes binding of libomc.so (include/omc.h).  No torch, no numpy compute: this module
only marshals arguments.  If the HIP library is missing or there is no GPU the calls fail
loudly -- there is no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

from . import _build

ABI_VERSION = 9

SEMANTICS = {"reference": 0, "textbook": 1, "two_pass": 2}
MODELS = {"gbm": 0, "heston": 1}
HESTON_SCHEMES = {"reference": 0, "clamp": 0, "full_truncation": 1, "calibrator": 2}


class OmcError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("model", C.c_int32), ("is_put", C.c_int32), ("semantics", C.c_int32),
                ("antithetic", C.c_int32), ("heston_scheme", C.c_int32), ("n_steps", C.c_int32),
                ("n_paths", C.c_int64),
                ("S0", C.c_double), ("K", C.c_double), ("r", C.c_double), ("sigma", C.c_double),
                ("T", C.c_double),
                ("v0", C.c_double), ("kappa", C.c_double), ("theta", C.c_double), ("xi", C.c_double),
                ("rho", C.c_double),
                ("seed", C.c_uint64), ("stream", C.c_uint64), ("pair_offset", C.c_uint64)]


class Result(C.Structure):
    _fields_ = [("price", C.c_double), ("sum", C.c_double), ("sumsq", C.c_double),
                ("std", C.c_double), ("zero_prob", C.c_double),
                ("n_paths", C.c_int64), ("n_exercised", C.c_int64), ("n_zero", C.c_int64),
                ("sum_nitm", C.c_int64),
                ("ms_paths", C.c_double), ("ms_lsm", C.c_double), ("ms_total", C.c_double),
                ("ms_pass1", C.c_double), ("ms_pass2", C.c_double), ("timed", C.c_int64),
                ("folded", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class MlpJob(C.Structure):
    """omc_mlp_job: one network of a batch trained side by side (omc_mlp_train_epoch_batch)."""
    _fields_ = [("data", C.c_void_p), ("n_rows", C.c_int64), ("batch", C.c_int64),
                ("params", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p),
                ("step", C.c_int64), ("lr", C.c_double), ("seed", C.c_uint64), ("shuffle_key", C.c_uint64),
                ("mean_loss", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)

# name -> (restype, argtypes); every symbol include/omc.h declares
_P, _I, _I64, _U64, _D, _SZ = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double, C.c_size_t
SIGNATURES = {
    "omc_abi_version": (C.c_int, []),
    "omc_last_error": (C.c_char_p, []),
    "omc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "omc_ctx_create": (C.c_int, [_I, _P, C.POINTER(_P)]),
    "omc_ctx_destroy": (C.c_int, [_P]),
    "omc_sync": (C.c_int, [_P]),
    "omc_alloc": (C.c_int, [_P, _SZ, C.POINTER(_P)]),
    "omc_free": (C.c_int, [_P, _P]),
    "omc_memcpy_h2d": (C.c_int, [_P, _P, _P, _SZ]),
    "omc_memcpy_d2h": (C.c_int, [_P, _P, _P, _SZ]),
    "omc_set_option": (C.c_int, [_P, C.c_char_p, _I64]),
    "omc_gbm_paths_f32": (C.c_int, [_P, _P, _I64, _I64, _I, _D, _D, _D, _D, _U64, _U64, _U64, _I]),
    "omc_heston_paths_f32": (C.c_int, [_P, _P, _I64, _I64, _I] + [_D] * 8 + [_U64, _U64, _U64, _I]),
    "omc_gbm_paths_from_normals_f32": (C.c_int, [_P, _P, _I64, _I64, _I, _D, _D, _D, _D, _P, _I64, _I]),
    "omc_heston_paths_from_normals_f32": (C.c_int, [_P, _P, _I64, _I64, _I] + [_D] * 8 + [_P, _P, _I64, _I]),
    "omc_philox4x32_10": (C.c_int, [_P, _P, _P, _I]),
    "omc_gbm_normals_f32": (C.c_int, [_P, _P, _I64, _I64, _I, _U64, _U64, _U64]),
    "omc_lsm_poly": (C.c_int, [_P, _P, _I64, _I64, _I, _D, _D, _D, _I, _I, C.POINTER(Result), _P, _P, _P]),
    "omc_lsm_apply_frozen": (C.c_int, [_P, _P, _I64, _I64, _I, _D, _D, _D, _I, _P, C.POINTER(Result), _P, _P]),
    "omc_lsm_apply_values": (C.c_int, [_P, _P, _I64, _I64, _I, _D, _D, _D, _I, _I, _P, _I64, C.POINTER(Result), _P, _P]),
    "omc_lsm_contnet": (C.c_int, [_P, _P, _I64, _I64, _I, _D, _D, _D, _I, _I, _I, _D, _U64, C.POINTER(Result), _P, _P]),
    "omc_price_american_contnet": (C.c_int, [_P, C.POINTER(Params), _I, _I, _D, _U64, C.POINTER(Result)]),
    "omc_contnet_init_params": (C.c_int, [_P, _I, _I, _U64, _P, _I]),
    "omc_set_allreduce_hook": (C.c_int, [_P, ALLREDUCE_FN, _P]),
    "omc_comm_unique_id": (C.c_int, [_P, _SZ]),
    "omc_comm_init": (C.c_int, [_P, _I, _I, _P, _SZ]),
    "omc_comm_destroy": (C.c_int, [_P]),
    "omc_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "omc_comm_allreduce_f64": (C.c_int, [_P, _P, _I, _I]),
    "omc_p2p_export": (C.c_int, [_P, _P, _SZ]),
    "omc_p2p_connect": (C.c_int, [_P, _I, _I, _P, _SZ]),
    "omc_p2p_disconnect": (C.c_int, [_P]),
    "omc_p2p_status": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "omc_price_american": (C.c_int, [_P, C.POINTER(Params), C.POINTER(Result), _P, _I64]),
    "omc_price_european": (C.c_int, [_P, C.POINTER(Params), C.POINTER(Result)]),
    "omc_heston_price_strikes": (C.c_int, [_P, _I64, _I] + [_D] * 8 + [_U64, _U64, _I, _P, _I, _I, _P, _P]),
    "omc_heston_price_surface": (C.c_int, [_P, _I64, _I] + [_D] * 7 + [_U64, _I, _P, _P, _I, _P, _P, _I, _I, _P, _P]),
    "omc_price_american_seq": (C.c_int, [_P, C.POINTER(Params), _I, C.POINTER(Result)]),
    "omc_seq_step_width": (C.c_int, [_P, C.POINTER(Params), _I]),
    "omc_price_american_batch": (C.c_int, [_P, C.POINTER(Params), _I, C.POINTER(Result)]),
    "omc_price_european_batch": (C.c_int, [_P, C.POINTER(Params), _I, C.POINTER(Result)]),
    "omc_price_american_contnet_batch": (C.c_int, [_P, C.POINTER(Params), _I, _I, _I, _D, C.POINTER(C.c_uint64),
                                                   C.POINTER(Result)]),
    "omc_mlp_param_count": (C.c_int, [_I, _I]),
    "omc_mlp_train_supported": (C.c_int, [_I, _I, _I64]),
    "omc_mlp_train_epoch": (C.c_int, [_P, _P, _I64, _I64, _I, _I, _P, _P, _P, C.POINTER(C.c_int64)]
                            + [_D] * 6 + [_U64, _U64, C.POINTER(C.c_double)]),
    "omc_mlp_train_batch_supported": (C.c_int, [_I, _I, _I64]),
    "omc_mlp_train_epoch_batch": (C.c_int, [_P, C.POINTER(MlpJob), _I, _I, _I] + [_D] * 5),
    "omc_mlp_shuffle_indices": (C.c_int, [_P, _I64, _U64, _P]),
    "omc_mlp_train_variant": (C.c_int, [C.c_int, C.c_int, _I64]),
    "omc_price_american_ols7": (C.c_int, [_P, C.POINTER(Params), C.POINTER(Result), _P, _P]),
    "omc_lsm_ols7": (C.c_int, [_P, _P, _I64, _I64, _I, _D, _D, _D, _I, C.POINTER(Result), _P, _P, _P, _P]),
    "omc_ctx_device_info": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int]),
    "omc_mlp_dropout_masks": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _I64, _P, C.c_uint32, _U64, C.c_double, _P]),
    "omc_nn_half_counts": (C.c_int, [_P, _P, _I64, _I64, _I, _D, _I, _P]),
    "omc_mlp_shard_epoch": (C.c_int, [_P, _P, _I64, _I64, _I64, _U64, _P, _P, _I, _I, _P, _P, _P]),
    "omc_mlp_train_epoch_sharded": (C.c_int, [_P, _P, _I64, _I64, _I64, _I, _I, _P, _P, _P, C.POINTER(C.c_int64)]
                                    + [_D] * 6 + [_U64, _P, _P, C.POINTER(C.c_double)]),
    "omc_lsm_apply_mlp": (C.c_int, [_P, _P, _I64, _I64, _I, _D, _D, _D, _I, _I, _I, _P, _P, _P, _D, _D, _D, _U64,
                                    C.POINTER(Result), _P, _P]),
    "omc_lsm_apply_mlp_shard": (C.c_int, [_P, _P, _I64, _I64, _I, _D, _D, _D, _I, _I, _I, _P, _P, _P, _D, _D, _D, _U64,
                                          C.POINTER(Result), _P, _P, _I64, _I64]),
    "omc_localvol_param_count": (C.c_int, [_I, _I]),
    "omc_localvol_paths_f32": (C.c_int, [_P, _P, _I64, _I64, _I, _D, _D, _D, _D, _I, _I, _P, _D, _D, _D, _P]),
    "omc_nn_build_rows": (C.c_int, [_P, _P, _I64, _I64, _I, _D, _D, _D, _I, _P, _I64, C.POINTER(C.c_int64), _P]),
    "omc_nn_feature_stats": (C.c_int, [_P, _P, _P, _P, _I64, _D, _D, _P]),
}

_lib = None
_lib_lock = threading.Lock()
_hip_runtime = None


def _preload_hip_runtime():
    """One process must use ONE HIP runtime.  The PyTorch-ROCm wheel bundles its own
    libamdhip64.so.7 / libhsa-runtime64 (torch/lib); if libomc.so pulled in /opt/rocm's copy
    first, a later `import torch` would find the GPU already owned by a different runtime and
    report no devices.  So when that wheel is installed, its runtime is dlopen'ed (RTLD_GLOBAL,
    without importing torch) before libomc.so, whose NEEDED libamdhip64.so.7 then resolves to
    the already-loaded copy.  OMC_HIP_RUNTIME=system forces /opt/rocm; =<path> forces a file."""
    global _hip_runtime
    if _hip_runtime is not None:
        return
    import importlib.util
    import sys
    mode = os.environ.get("OMC_HIP_RUNTIME", "auto")
    if mode == "system" or "torch" in sys.modules:
        return
    cand = None
    if mode not in ("auto", "torch") and os.path.exists(mode):
        cand = mode
    else:
        try:
            spec = importlib.util.find_spec("torch")
        except (ImportError, ValueError):
            spec = None
        if spec is not None and spec.submodule_search_locations:
            p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
            if os.path.exists(p):
                cand = p
    if cand:
        try:
            _hip_runtime = C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            _hip_runtime = None


def load_library(build_if_missing: bool = True):
    """dlopen libomc.so (building it in-tree first if needed).  No HIP call happens here."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        path = _build.LIB
        if not os.path.exists(path) or (build_if_missing and _build._stale()):
            if not build_if_missing:
                raise OmcError(f"{path} is missing: run `python -m options_model_amd._build`")
            try:
                path = _build.build()
            except Exception as e:  # stale-but-present library is still usable on a box w/o hipcc
                if not os.path.exists(_build.LIB):
                    raise OmcError(f"libomc.so is missing and cannot be built: {e}") from e
                path = _build.LIB
        _preload_hip_runtime()
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError = symbol missing = broken build
            fn.restype = res
            fn.argtypes = args
        if lib.omc_abi_version() != ABI_VERSION:
            raise OmcError("libomc.so ABI version mismatch: rebuild the library")
        _lib = lib
        return lib


def _check(lib, rc):
    if rc == 0:
        return
    msg = (lib.omc_last_error() or b"").decode(errors="replace")
    if rc < 0:
        raise ValueError(msg)
    raise OmcError(f"HIP error {rc}: {msg}")


def comm_unique_id() -> bytes:
    """Rank 0: a fresh RCCL unique id (128 bytes) for Context.comm_init on every rank."""
    lib = load_library()
    buf = C.create_string_buffer(128)
    _check(lib, lib.omc_comm_unique_id(buf, 128))
    return buf.raw


def device_count() -> int:
    lib = load_library()
    n = C.c_int(0)
    rc = lib.omc_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class DeviceArray:
    """A device allocation owned by a Context (freed with it or explicitly)."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        _check(ctx.lib, ctx.lib.omc_alloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value
        ctx._arrays.add(self)

    def to_host(self):
        out = np.empty(self.shape, self.dtype)
        _check(self.ctx.lib, self.ctx.lib.omc_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr,
                                                          self.nbytes))
        return out

    def from_host(self, a):
        a = np.ascontiguousarray(a, self.dtype)
        assert a.shape == self.shape, (a.shape, self.shape)
        _check(self.ctx.lib, self.ctx.lib.omc_memcpy_h2d(self.ctx.handle, self.ptr, a.ctypes.data,
                                                          self.nbytes))
        return self

    def free(self):
        if self.ptr:
            self.ctx.lib.omc_free(self.ctx.handle, self.ptr)
            self.ptr = None
            self.ctx._arrays.discard(self)


class Context:
    """One per (process, device).  Calls on a context are serialised by the caller."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib, self.lib.omc_ctx_create(int(device), C.c_void_p(stream) if stream else None,
                                                  C.byref(h)))
        self.handle = h
        self.device = device
        self._arrays = set()
        self._hook = None

    # -- housekeeping
    def close(self):
        if self.handle:
            for a in list(self._arrays):
                a.free()
            self.lib.omc_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def empty(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype)

    def to_device(self, a, dtype=None):
        a = np.ascontiguousarray(a, dtype or a.dtype)
        return DeviceArray(self, a.shape, a.dtype).from_host(a)

    def sync(self):
        _check(self.lib, self.lib.omc_sync(self.handle))

    def set_option(self, key: str, value: int):
        _check(self.lib, self.lib.omc_set_option(self.handle, key.encode(), int(value)))

    def set_allreduce_hook(self, fn):
        """fn(dptr:int, count:int) -> None must sum-all-reduce `count` device doubles in place."""
        if fn is None:
            self._hook = None
            _check(self.lib, self.lib.omc_set_allreduce_hook(self.handle, C.cast(None, ALLREDUCE_FN), None))
            return

        def tramp(_user, dptr, count):
            try:
                fn(dptr, count)
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        self._hook = ALLREDUCE_FN(tramp)
        _check(self.lib, self.lib.omc_set_allreduce_hook(self.handle, self._hook, None))

    # -- RNG taps
    def philox4x32_10(self, ctr_key):
        a = np.ascontiguousarray(ctr_key, np.uint32).reshape(-1, 6)
        out = np.empty((a.shape[0], 4), np.uint32)
        _check(self.lib, self.lib.omc_philox4x32_10(self.handle, a.ctypes.data, out.ctypes.data,
                                                     a.shape[0]))
        return out

    def gbm_normals(self, n_pairs, n_steps, seed, stream=0, pair_offset=0):
        Z = self.empty((n_steps, n_pairs), np.float32)
        _check(self.lib, self.lib.omc_gbm_normals_f32(self.handle, Z.ptr, n_pairs, n_pairs, n_steps,
                                                       seed, stream, pair_offset))
        return Z

    # -- paths (device matrices [n_steps+1][n_paths], ld == n_paths)
    def gbm_paths(self, n_paths, n_steps, S0, r, sigma, T, seed, stream=0, pair_offset=0,
                  antithetic=True, out=None):
        S = out if out is not None else self.empty((n_steps + 1, n_paths), np.float32)
        _check(self.lib, self.lib.omc_gbm_paths_f32(self.handle, S.ptr, S.shape[1], n_paths, n_steps,
                                                     S0, r, sigma, T, seed, stream, pair_offset,
                                                     int(antithetic)))
        return S

    def heston_paths(self, n_paths, n_steps, S0, r, T, v0, kappa, theta, xi, rho, seed, stream=0,
                     pair_offset=0, scheme=0, out=None):
        S = out if out is not None else self.empty((n_steps + 1, n_paths), np.float32)
        _check(self.lib, self.lib.omc_heston_paths_f32(self.handle, S.ptr, S.shape[1], n_paths,
                                                        n_steps, S0, r, T, v0, kappa, theta, xi, rho,
                                                        seed, stream, pair_offset, scheme))
        return S

    def gbm_paths_from_normals(self, z_half, S0, r, sigma, T, antithetic=True):
        z = self.to_device(z_half, np.float32)
        N, P = z.shape
        M = 2 * P if antithetic else P
        S = self.empty((N + 1, M), np.float32)
        _check(self.lib, self.lib.omc_gbm_paths_from_normals_f32(self.handle, S.ptr, M, M, N, S0, r,
                                                                  sigma, T, z.ptr, P, int(antithetic)))
        z.free()
        return S

    def heston_paths_from_normals(self, z1_half, z2_half, S0, r, T, v0, kappa, theta, xi, rho,
                                  scheme=0):
        z1 = self.to_device(z1_half, np.float32)
        z2 = self.to_device(z2_half, np.float32)
        N, P = z1.shape
        S = self.empty((N + 1, 2 * P), np.float32)
        _check(self.lib, self.lib.omc_heston_paths_from_normals_f32(
            self.handle, S.ptr, 2 * P, 2 * P, N, S0, r, T, v0, kappa, theta, xi, rho, z1.ptr, z2.ptr,
            P, scheme))
        z1.free()
        z2.free()
        return S

    # -- backward induction on a device path matrix
    def lsm_poly(self, S, K, r, T, is_put, semantics="reference", want_state=False, n_paths=None):
        N = S.shape[0] - 1
        ld = S.shape[1]
        M = ld if n_paths is None else n_paths
        res = Result()
        betas = np.zeros((N + 1, 4))
        sx = np.zeros(M, np.float32) if want_state else None
        tex = np.zeros(M, np.int32) if want_state else None
        ptr = S.ptr if isinstance(S, DeviceArray) else int(S.data_ptr())
        _check(self.lib, self.lib.omc_lsm_poly(self.handle, ptr, ld, M, N, K, r, T, int(is_put),
                                                SEMANTICS[semantics], C.byref(res), betas.ctypes.data,
                                                sx.ctypes.data if want_state else None,
                                                tex.ctypes.data if want_state else None))
        d = res.as_dict()
        d.update(betas=betas[:, :3], nitm=betas[:, 3].astype(np.int64), sx=sx, tex=tex)
        return d

    def lsm_ols7(self, S, K, r, T, is_put, want_state=False, n_paths=None):
        """Two-pass flow with the global least-squares fit on the reference's 7 features (omc_lsm_ols7) on a device path
        matrix -> result dict + weights [7], feat_mean / feat_std [7], y_mean, y_std (+ sx, tex if want_state)."""
        N = S.shape[0] - 1
        ld = S.shape[1]
        M = ld if n_paths is None else n_paths
        res = Result()
        w = np.zeros(7)
        st = np.zeros(16)
        sx = np.zeros(M, np.float32) if want_state else None
        tex = np.zeros(M, np.int32) if want_state else None
        ptr = S.ptr if isinstance(S, DeviceArray) else int(S.data_ptr())
        _check(self.lib, self.lib.omc_lsm_ols7(self.handle, ptr, ld, M, N, K, r, T, int(is_put), C.byref(res), w.ctypes.data,
                                                st.ctypes.data, sx.ctypes.data if want_state else None,
                                                tex.ctypes.data if want_state else None))
        d = res.as_dict()
        d.update(weights=w, feat_mean=st[:7].copy(), feat_std=st[7:14].copy(), y_mean=float(st[14]), y_std=float(st[15]),
                 sx=sx, tex=tex)
        return d

    def lsm_apply_frozen(self, S, K, r, T, is_put, betas4, want_state=True):
        N, M = S.shape[0] - 1, S.shape[1]
        b = np.ascontiguousarray(betas4, np.float64)
        assert b.shape == (N + 1, 4)
        res = Result()
        sx = np.zeros(M, np.float32) if want_state else None
        tex = np.zeros(M, np.int32) if want_state else None
        _check(self.lib, self.lib.omc_lsm_apply_frozen(self.handle, S.ptr, M, M, N, K, r, T,
                                                        int(is_put), b.ctypes.data, C.byref(res),
                                                        sx.ctypes.data if want_state else None,
                                                        tex.ctypes.data if want_state else None))
        d = res.as_dict()
        d.update(sx=sx, tex=tex)
        return d

    def lsm_apply_values(self, S, K, r, T, is_put, cont, semantics="reference", want_state=True):
        """Per-step sweep driven by a device float32 matrix of continuation values [N+1][M]."""
        N, M = S.shape[0] - 1, S.shape[1]
        assert cont.shape == (N + 1, M) and cont.dtype == np.float32
        res = Result()
        sx = np.zeros(M, np.float32) if want_state else None
        tex = np.zeros(M, np.int32) if want_state else None
        _check(self.lib, self.lib.omc_lsm_apply_values(self.handle, S.ptr, M, M, N, K, r, T, int(is_put),
                                                        SEMANTICS[semantics], cont.ptr, M, C.byref(res),
                                                        sx.ctypes.data if want_state else None,
                                                        tex.ctypes.data if want_state else None))
        d = res.as_dict()
        d.update(sx=sx, tex=tex)
        return d

    def lsm_contnet(self, S, K, r, T, is_put, nn_hidden=32, nn_epochs=10, nn_lr=1e-3, nn_seed=0, want_state=True):
        """The reference's v1 / v2 per-step flow (fresh ContNet per step) on a device path matrix."""
        N, M = S.shape[0] - 1, S.shape[1]
        res = Result()
        sx = np.zeros(M, np.float32) if want_state else None
        tex = np.zeros(M, np.int32) if want_state else None
        _check(self.lib, self.lib.omc_lsm_contnet(self.handle, S.ptr, M, M, N, K, r, T, int(is_put), int(nn_hidden),
                                                   int(nn_epochs), float(nn_lr), int(nn_seed) & (2**64 - 1),
                                                   C.byref(res), sx.ctypes.data if want_state else None,
                                                   tex.ctypes.data if want_state else None))
        d = res.as_dict()
        d.update(sx=sx, tex=tex)
        return d

    def contnet_init_params(self, nn_hidden, t, nn_seed=0):
        """Initial parameters of step t's net as a dict of torch-shaped arrays (w0 [h,1], b0, w1 [h,h], b1, w2 [1,h], b2)."""
        H = 32 if nn_hidden <= 32 else 64 if nn_hidden <= 64 else 128
        n = 8 * H + H * H + 2 * H + 1
        flat = np.zeros(n, np.float32)
        _check(self.lib, self.lib.omc_contnet_init_params(self.handle, int(nn_hidden), int(t),
                                                           int(nn_seed) & (2**64 - 1), flat.ctypes.data, n))
        h = int(nn_hidden)
        l0 = flat[:8 * H].reshape(H, 8)
        w1 = flat[8 * H:8 * H + H * H].reshape(H, H)
        b1 = flat[8 * H + H * H:8 * H + H * H + H]
        wo = flat[8 * H + H * H + H:8 * H + H * H + 2 * H]
        return dict(w0=l0[:h, :1].copy(), b0=l0[:h, 7].copy(), w1=w1[:h, :h].copy(), b1=b1[:h].copy(),
                    w2=wo[:h].reshape(1, h).copy(), b2=flat[-1:].copy(), flat=flat)

    def price_american_contnet(self, params: Params, nn_hidden=32, nn_epochs=10, nn_lr=1e-3, nn_seed=0):
        res = Result()
        _check(self.lib, self.lib.omc_price_american_contnet(self.handle, C.byref(params), int(nn_hidden),
                                                              int(nn_epochs), float(nn_lr),
                                                              int(nn_seed) & (2**64 - 1), C.byref(res)))
        return res.as_dict()

    # -- native RCCL communicator (no torch): see include/omc.h
    def comm_init(self, rank: int, world: int, uid: bytes):
        buf = C.create_string_buffer(bytes(uid), 128)
        _check(self.lib, self.lib.omc_comm_init(self.handle, int(rank), int(world), buf, 128))

    def comm_destroy(self):
        _check(self.lib, self.lib.omc_comm_destroy(self.handle))

    def comm_info(self):
        """-> (rank, world) of the live communicator; world == 0: none."""
        r, w = C.c_int(0), C.c_int(0)
        _check(self.lib, self.lib.omc_comm_info(self.handle, C.byref(r), C.byref(w)))
        return r.value, w.value

    def device_info(self) -> dict:
        """-> dict(device=HIP ordinal, pci_bus_id="0000:c1:00.0", name=...) of the card this context runs on."""
        dev = C.c_int(-1)
        pci, name = C.create_string_buffer(64), C.create_string_buffer(256)
        _check(self.lib, self.lib.omc_ctx_device_info(self.handle, C.byref(dev), pci, 64, name, 256))
        return dict(device=dev.value, pci_bus_id=pci.value.decode(), name=name.value.decode())

    def comm_allreduce(self, values, op="sum"):
        a = np.ascontiguousarray(values, np.float64).copy()
        _check(self.lib, self.lib.omc_comm_allreduce_f64(self.handle, a.ctypes.data, a.size, 1 if op == "max" else 0))
        return a

    # -- direct peer exchange of the per-step moments (include/omc.h)
    def p2p_export(self) -> bytes:
        buf = C.create_string_buffer(64)
        _check(self.lib, self.lib.omc_p2p_export(self.handle, buf, 64))
        return buf.raw

    def p2p_connect(self, rank: int, world: int, handles: bytes):
        buf = C.create_string_buffer(bytes(handles), 64 * int(world))
        _check(self.lib, self.lib.omc_p2p_connect(self.handle, int(rank), int(world), buf, 64 * int(world)))

    def p2p_disconnect(self):
        _check(self.lib, self.lib.omc_p2p_disconnect(self.handle))

    def p2p_status(self):
        """-> (connected, world, sticky error word)"""
        a, b, e = C.c_int(0), C.c_int(0), C.c_uint64(0)
        _check(self.lib, self.lib.omc_p2p_status(self.handle, C.byref(a), C.byref(b), C.byref(e)))
        return bool(a.value), b.value, e.value

    # -- fused pricing
    def price_american(self, params: Params, keep_paths: DeviceArray | None = None):
        res = Result()
        _check(self.lib, self.lib.omc_price_american(
            self.handle, C.byref(params), C.byref(res), keep_paths.ptr if keep_paths else None,
            keep_paths.shape[1] if keep_paths else 0))
        return res.as_dict()

    def price_american_ols7(self, params: Params):
        """Fused pricing with regressor "ols7" (paths + omc_lsm_ols7 in the context's own matrix) -> result dict + fit."""
        res = Result()
        w, st = np.zeros(7), np.zeros(16)
        _check(self.lib, self.lib.omc_price_american_ols7(self.handle, C.byref(params), C.byref(res), w.ctypes.data,
                                                           st.ctypes.data))
        d = res.as_dict()
        d.update(weights=w, feat_mean=st[:7].copy(), feat_std=st[7:14].copy(), y_mean=float(st[14]), y_std=float(st[15]))
        return d

    def price_european(self, params: Params):
        res = Result()
        _check(self.lib, self.lib.omc_price_european(self.handle, C.byref(params), C.byref(res)))
        return res.as_dict()

    def heston_price_strikes(self, n_paths, n_steps, S0, r, T, v0, kappa, theta, xi, rho, strikes,
                             is_put=False, seed=42, stream=0, scheme=2):
        """One expiry, many strikes (the calibrator's inner call) -> (prices, stderrs)."""
        k = np.ascontiguousarray(strikes, np.float64)
        prices = np.empty_like(k)
        errs = np.empty_like(k)
        _check(self.lib, self.lib.omc_heston_price_strikes(
            self.handle, int(n_paths), int(n_steps), S0, r, T, v0, kappa, theta, xi, rho, int(seed),
            int(stream), int(scheme), k.ctypes.data, k.size, int(is_put), prices.ctypes.data,
            errs.ctypes.data))
        return prices, errs

    def heston_price_surface(self, n_paths, n_steps, S0, r, v0, kappa, theta, xi, rho, expiries, streams, strikes,
                             expiry_of, is_put=False, seed=42, scheme=2):
        """Many expiries x many strikes in one launch set (one objective evaluation of the calibrator): quote q = strike
        strikes[q] on expiry expiries[expiry_of[q]], simulated on Philox sub-stream streams[expiry_of[q]]
        -> (prices, stderrs), each quote bit-equal to its own heston_price_strikes call."""
        T = np.ascontiguousarray(expiries, np.float64)
        st = np.ascontiguousarray(streams, np.uint64)
        k = np.ascontiguousarray(strikes, np.float64)
        eo = np.ascontiguousarray(expiry_of, np.int32)
        if st.shape != T.shape or eo.shape != k.shape or T.ndim != 1 or k.ndim != 1:
            raise ValueError("expiries / streams and strikes / expiry_of must be matching 1-d arrays")
        prices = np.empty_like(k)
        errs = np.empty_like(k)
        _check(self.lib, self.lib.omc_heston_price_surface(
            self.handle, int(n_paths), int(n_steps), S0, r, v0, kappa, theta, xi, rho, int(seed), int(scheme),
            T.ctypes.data, st.ctypes.data, T.size, k.ctypes.data, eo.ctypes.data, k.size, int(is_put), prices.ctypes.data,
            errs.ctypes.data))
        return prices, errs

    def mlp_train_epoch(self, data_ptr, n_rows, batch, params_ptr, m_ptr, v_ptr, step, lr, dropout, seed,
                        hidden=64, layers=2, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-5,
                        shuffle_key=0):
        """One epoch of the fused 7->64->64->1 trainer on device pointers -> (mean loss, new step).
        shuffle_key != 0: rows are visited in the keyed pseudo-random permutation."""
        st = C.c_int64(int(step))
        loss = C.c_double(0.0)
        _check(self.lib, self.lib.omc_mlp_train_epoch(
            self.handle, int(data_ptr), int(n_rows), int(batch), int(hidden), int(layers), int(params_ptr),
            int(m_ptr), int(v_ptr), C.byref(st), float(lr), float(beta1), float(beta2), float(eps),
            float(weight_decay), float(dropout), int(seed), int(shuffle_key), C.byref(loss)))
        return loss.value, st.value

    # -- the NN regressor sharded over ranks (include/omc.h; host logic in nn_dist.py)
    def nn_half_counts(self, S_ptr, ld, n_paths, n_steps, K, is_put):
        """-> int64 [(n_steps - 1), 2]: in-the-money paths per step (index n_steps - 1 - t) in the first / second half
        of the rank's columns."""
        out = np.zeros((max(int(n_steps) - 1, 0), 2), np.int64)
        _check(self.lib, self.lib.omc_nn_half_counts(self.handle, int(S_ptr), int(ld), int(n_paths), int(n_steps), float(K),
                                                     int(bool(is_put)), out.ctypes.data))
        return out

    def mlp_shard_epoch(self, data_ptr, n_rows_local, rows_global, batch, shuffle_key, gstart, lstart, data_epoch_ptr,
                        drop_pos_ptr, segs_per_step=0):
        """This rank's rows of the epoch, gathered in epoch order -> step_off (int64 [steps + 1], host)."""
        g = np.ascontiguousarray(gstart, np.int64)
        l_ = np.ascontiguousarray(lstart, np.int64)
        assert g.size == l_.size + 1
        steps = (int(rows_global) + int(batch) - 1) // int(batch)
        so = np.zeros(steps + 1, np.int64)
        _check(self.lib, self.lib.omc_mlp_shard_epoch(
            self.handle, int(data_ptr) if n_rows_local else None, int(n_rows_local), int(rows_global), int(batch),
            int(shuffle_key), g.ctypes.data, l_.ctypes.data, int(l_.size), int(segs_per_step),
            int(data_epoch_ptr) if n_rows_local else None,
            int(drop_pos_ptr) if n_rows_local else None, so.ctypes.data))
        return so

    def mlp_train_epoch_sharded(self, data_epoch_ptr, n_rows_local, rows_global, batch, params_ptr, m_ptr, v_ptr, step, lr,
                                dropout, seed, step_off, drop_pos_ptr, hidden=64, layers=2, beta1=0.9, beta2=0.999,
                                eps=1e-8, weight_decay=1e-5):
        """One epoch of the job's training on this rank's rows -> (the job's mean loss, new step)."""
        st = C.c_int64(int(step))
        loss = C.c_double(0.0)
        so = np.ascontiguousarray(step_off, np.int64)
        _check(self.lib, self.lib.omc_mlp_train_epoch_sharded(
            self.handle, int(data_epoch_ptr) if n_rows_local else None, int(n_rows_local), int(rows_global), int(batch),
            int(hidden), int(layers), int(params_ptr), int(m_ptr), int(v_ptr), C.byref(st), float(lr), float(beta1),
            float(beta2), float(eps), float(weight_decay), float(dropout), int(seed), so.ctypes.data,
            int(drop_pos_ptr) if n_rows_local else None, C.byref(loss)))
        return loss.value, st.value

    def mlp_train_epoch_batch(self, jobs, hidden, layers, dropout, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-5):
        """One epoch for every network of `jobs` (list of dicts: data_ptr, n_rows, batch, params_ptr, m_ptr, v_ptr,
        step, lr, seed, shuffle_key), all of shape hidden x layers, side by side -> [(mean_loss, new_step), ...]."""
        n = len(jobs)
        arr = (MlpJob * n)()
        for a, j in zip(arr, jobs):
            a.data, a.n_rows, a.batch = int(j["data_ptr"]), int(j["n_rows"]), int(j["batch"])
            a.params, a.adam_m, a.adam_v = int(j["params_ptr"]), int(j["m_ptr"]), int(j["v_ptr"])
            a.step, a.lr = int(j["step"]), float(j["lr"])
            a.seed, a.shuffle_key = int(j["seed"]) & (2**64 - 1), int(j["shuffle_key"]) & (2**64 - 1)
        _check(self.lib, self.lib.omc_mlp_train_epoch_batch(self.handle, arr, n, int(hidden), int(layers), float(beta1),
                                                            float(beta2), float(eps), float(weight_decay), float(dropout)))
        return [(a.mean_loss, a.step) for a in arr]

    def lsm_apply_mlp(self, S_ptr, ld, n_paths, n_steps, K, r, T, is_put, params_ptr, feat_mean, feat_std,
                      y_mean, y_std, dropout, seed, want_state=False, hidden=64, layers=2, col_bases=None):
        """Pass 2 of the NN flow on a device path matrix -> result dict (+ sx, tex if want_state).
        col_bases = (base0, base1): the matrix is one rank's shard (omc_lsm_apply_mlp_shard)."""
        fm = np.ascontiguousarray(feat_mean, np.float64)
        fs = np.ascontiguousarray(feat_std, np.float64)
        assert fm.size == 7 and fs.size == 7
        res = Result()
        sx = np.empty(n_paths, np.float32) if want_state else None
        tex = np.empty(n_paths, np.int32) if want_state else None
        b0, b1 = (0, int(n_paths) // 2) if col_bases is None else (int(col_bases[0]), int(col_bases[1]))
        _check(self.lib, self.lib.omc_lsm_apply_mlp_shard(
            self.handle, int(S_ptr), int(ld), int(n_paths), int(n_steps), float(K), float(r), float(T),
            int(bool(is_put)), int(hidden), int(layers), int(params_ptr), fm.ctypes.data, fs.ctypes.data,
            float(y_mean), float(y_std), float(dropout), int(seed), C.byref(res),
            sx.ctypes.data if want_state else None, tex.ctypes.data if want_state else None, b0, b1))
        out = res.as_dict()
        if want_state:
            out["sx"], out["tex"] = sx, tex
        return out

    def nn_build_rows(self, S_ptr, ld, n_paths, n_steps, K, r, T, is_put, data_ptr=None, cap_rows=0):
        """Pass 1 of the NN flow from a device path matrix.  data_ptr None -> row count only;
        else -> (rows, feat_mean[7], feat_std[7], y_mean, y_std) with the rows written to data_ptr."""
        n = C.c_int64(0)
        st = np.zeros(16)
        _check(self.lib, self.lib.omc_nn_build_rows(
            self.handle, int(S_ptr), int(ld), int(n_paths), int(n_steps), float(K), float(r), float(T),
            int(bool(is_put)), int(data_ptr) if data_ptr else None, int(cap_rows), C.byref(n), st.ctypes.data))
        if not data_ptr:
            return n.value
        return n.value, st[:7].copy(), st[7:14].copy(), float(st[14]), float(st[15])

    def nn_feature_stats(self, x_ptr, t_ptr, y_ptr, n_rows, T, dt):
        """-> (means[7], variances[7]) of [x, x^2, x^3, max(x-1,0), s, x*s, y] over device rows."""
        out = np.zeros(16)
        _check(self.lib, self.lib.omc_nn_feature_stats(self.handle, int(x_ptr), int(t_ptr), int(y_ptr),
                                                       int(n_rows), float(T), float(dt), out.ctypes.data))
        return out[:7].copy(), out[8:15].copy()

    def mlp_dropout_masks(self, variant, hidden, layers, n_rows, step, seed, dropout, keys=None):
        """keep (True) / drop of every hidden activation as kernel `variant` draws it -> bool [layers, n_rows, hidden]
        (variant 0: pass 2, keys = path columns, step = time step; 1 .. 4: omc_mlp_train_variant)."""
        out = np.zeros((int(layers), int(n_rows), int(hidden)), np.uint8)
        k = None if keys is None else np.ascontiguousarray(keys, np.uint32)
        assert k is None or k.size == n_rows
        _check(self.lib, self.lib.omc_mlp_dropout_masks(self.handle, int(variant), int(hidden), int(layers), int(n_rows),
                                                         k.ctypes.data if k is not None else None, int(step),
                                                         int(seed) & (2 ** 64 - 1), float(dropout), out.ctypes.data))
        return out.astype(bool)

    def mlp_shuffle_indices(self, n_rows, shuffle_key, out_ptr):
        _check(self.lib, self.lib.omc_mlp_shuffle_indices(self.handle, int(n_rows), int(shuffle_key), int(out_ptr)))

    # -- many small pricings as one set of launches (all share model/semantics/antithetic)
    MAX_BATCH = 65535

    def _batch(self, fn, params_list):
        out = []
        for lo in range(0, len(params_list), self.MAX_BATCH):
            chunk = params_list[lo:lo + self.MAX_BATCH]
            n = len(chunk)
            arr = (Params * n)(*chunk)
            res = (Result * n)()
            _check(self.lib, fn(self.handle, arr, n, res))
            out.extend(r.as_dict() for r in res)
        return out

    def price_american_seq(self, params_list):
        """n pricings back to back on the stream, one wait at the end -> list of result dicts."""
        plist = list(params_list)
        n = len(plist)
        arr = (Params * n)(*plist)
        res = (Result * n)()
        _check(self.lib, self.lib.omc_price_american_seq(self.handle, arr, n, res))
        return [r.as_dict() for r in res]

    def seq_step_width(self, params_list) -> int:
        """How many pricings of this sequence share one launch per time step (per-step flows; 1 = none)."""
        plist = list(params_list)
        arr = (Params * len(plist))(*plist)
        return int(self.lib.omc_seq_step_width(self.handle, arr, len(plist)))

    def price_american_batch(self, params_list):
        return self._batch(self.lib.omc_price_american_batch, list(params_list))

    def price_american_contnet_batch(self, params_list, nn_hidden=32, nn_epochs=10, nn_lr=1e-3, nn_seeds=0):
        """Many pricings with the v1 / v2 regressor (fresh ContNet per step) as one set of launches.
        nn_seeds: one seed for all, or one per problem."""
        plist = list(params_list)
        n = len(plist)
        seeds = [int(nn_seeds)] * n if isinstance(nn_seeds, (int, np.integer)) else [int(x) for x in nn_seeds]
        assert len(seeds) == n
        out = []
        for lo in range(0, n, self.MAX_BATCH):
            chunk = plist[lo:lo + self.MAX_BATCH]
            k = len(chunk)
            arr = (Params * k)(*chunk)
            sd = (C.c_uint64 * k)(*[x & (2**64 - 1) for x in seeds[lo:lo + k]])
            res = (Result * k)()
            _check(self.lib, self.lib.omc_price_american_contnet_batch(self.handle, arr, k, int(nn_hidden), int(nn_epochs),
                                                                       float(nn_lr), sd, res))
            out.extend(r.as_dict() for r in res)
        return out

    def price_european_batch(self, params_list):
        return self._batch(self.lib.omc_price_european_batch, list(params_list))


def make_params(model="gbm", is_put=True, semantics="reference", antithetic=True,
                heston_scheme="reference", n_paths=0, n_steps=0, S0=100.0, K=100.0, r=0.05,
                sigma=0.2, T=1.0, v0=0.04, kappa=2.0, theta=0.04, xi=0.3, rho=-0.7, seed=42,
                stream=0, pair_offset=0) -> Params:
    p = Params()
    p.model = MODELS[model.lower()]
    p.is_put = int(bool(is_put))
    p.semantics = SEMANTICS[semantics]
    p.antithetic = int(bool(antithetic))
    p.heston_scheme = HESTON_SCHEMES[heston_scheme] if isinstance(heston_scheme, str) else int(heston_scheme)
    p.n_steps = int(n_steps)
    p.n_paths = int(n_paths)
    p.S0, p.K, p.r, p.sigma, p.T = float(S0), float(K), float(r), float(sigma or 0.0), float(T)
    p.v0, p.kappa, p.theta, p.xi, p.rho = float(v0), float(kappa), float(theta), float(xi), float(rho)
    p.seed, p.stream, p.pair_offset = int(seed), int(stream), int(pair_offset)
    return p


_default_ctx = {}
_ctx_lock = threading.Lock()


def resolve_device(device=None) -> int:
    """Which HIP device a call without an explicit `device` uses.  Default 0.  OMC_DEVICE=<k> picks device k;
    OMC_DEVICE=auto spreads PROCESSES over the node's GPUs by pid -- the reference's UIs fan the spot values of a curve
    job over a spawn pool (options_model_2_ui.py:87-133, options_model_3.py:1043-1056): with `auto` those workers,
    whose pids are consecutive, land on different GPUs without a change to the caller."""
    if device is not None:
        return int(device)
    env = os.environ.get("OMC_DEVICE", "").strip().lower()
    if not env:
        return 0
    if env == "auto":
        n = device_count()
        return os.getpid() % n if n > 0 else 0
    return int(env)


def default_context(device=None) -> Context:
    """Lazily created per (process, device) -- safe under spawn'ed worker pools."""
    device = resolve_device(device)
    key = (os.getpid(), device)
    with _ctx_lock:
        ctx = _default_ctx.get(key)
        if ctx is None:
            ctx = Context(device)
            _default_ctx[key] = ctx
        return ctx


_pools = {}


def context_pool(n: int, device: int = 0) -> list:
    """n DEDICATED contexts on one device (own stream and workspaces each; never default_context(), which other
    code of the process may be using at the same time -- an omc_ctx is not thread-safe): for work made of many
    small, latency-bound pricings -- the per-step network flow of the v1 / v2 curves -- which overlap on the GPU
    when issued from several host threads (ctypes releases the GIL during a call)."""
    key = (os.getpid(), device)
    with _ctx_lock:
        pool = _pools.setdefault(key, [])
        while len(pool) < n:
            pool.append((Context(device), threading.Lock()))
        return pool[:n]


def map_contexts(fn, items, workers: int | None = None, device: int = 0) -> list:
    """[fn(ctx, item) for item in items], in order, run by `workers` host threads with one context each
    (OMC_CURVE_STREAMS, default 8; 1 = plain loop on the default context).  Every pool context is used under
    its own lock, so two map_contexts calls running at once (two threads of the caller) share the pool safely."""
    items = list(items)
    if workers is None:
        workers = int(os.environ.get("OMC_CURVE_STREAMS", "8"))
    workers = max(1, min(int(workers), len(items)))
    if workers == 1:
        ctx = default_context(device)
        return [fn(ctx, it) for it in items]
    from concurrent.futures import ThreadPoolExecutor
    pool = context_pool(workers, device)

    def one(args):
        i, it = args
        ctx, lock = pool[i % workers]
        with lock:
            return fn(ctx, it)

    with ThreadPoolExecutor(max_workers=workers) as ex:
        return list(ex.map(one, enumerate(items)))
