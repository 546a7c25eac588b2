"""ctypes loader for oracle/_build/libomc_oracle.so.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libomc_oracle.so")
_lib = None


class LsmResult(C.Structure):
    _fields_ = [("price", C.c_double), ("sum", C.c_double), ("sumsq", C.c_double),
                ("n_paths", C.c_int64), ("n_exercised", C.c_int64), ("n_zero", C.c_int64),
                ("sum_nitm", C.c_int64)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "omc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        f32p, i64, i32, u64, u32, dbl = (C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_uint32,
                                          C.c_double)
        _lib.orc_philox4x32_10.argtypes = [C.c_void_p] * 3
        _lib.orc_gbm_normals_f32.argtypes = [f32p, i64, i64, i32, u64, u32, u64]
        _lib.orc_gbm_paths_f32.argtypes = [f32p, i64, i64, i32, dbl, dbl, dbl, dbl, u64, u32, u64, i32]
        _lib.orc_gbm_paths_from_normals_f32.argtypes = [f32p, i64, i64, i32, dbl, dbl, dbl, dbl,
                                                        f32p, i64, i32]
        _lib.orc_heston_paths_f32.argtypes = [f32p, i64, i64, i32] + [dbl] * 8 + [u64, u32, u64, i32]
        _lib.orc_heston_paths_from_normals_f32.argtypes = [f32p, i64, i64, i32] + [dbl] * 8 + [
            f32p, f32p, i64, i32]
        _lib.orc_lsm_poly.argtypes = [f32p, i64, i64, i32, dbl, dbl, dbl, i32, i32,
                                      C.POINTER(LsmResult), C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]
        _lib.orc_lsm_poly.restype = C.c_int
        _lib.orc_lsm_apply_frozen.argtypes = [f32p, i64, i64, i32, dbl, dbl, dbl, i32, C.c_void_p,
                                              C.c_void_p, C.POINTER(LsmResult), C.c_void_p,
                                              C.c_void_p]
        _lib.orc_lsm_apply_frozen.restype = C.c_int
        _lib.orc_lsm_pass1_moments.argtypes = [f32p, i64, i64, i32, dbl, dbl, dbl, i32, C.c_void_p]
        _lib.orc_fold_table.argtypes = [C.c_void_p, i32, dbl, dbl]
        _lib.orc_fold_constants.argtypes = [dbl, dbl, dbl, dbl, dbl, i32, C.c_void_p, C.c_void_p]
        _lib.orc_lsm_two_pass_folded.argtypes = [f32p, i64, i64, i32, dbl, dbl, dbl, i32, dbl, dbl,
                                                 C.POINTER(LsmResult), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_lsm_two_pass_folded.restype = C.c_int
        _lib.orc_solve_poly2.argtypes = [C.c_void_p, C.c_void_p]
        _lib.orc_heston_terminal_f32.argtypes = [f32p, i64, i32] + [dbl] * 8 + [u64, u32, u64, i32]
        _lib.orc_european_from_paths.argtypes = [f32p, i64, i64, i32, dbl, dbl, dbl, i32,
                                                 C.c_void_p, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def philox4x32_10(ctr, key):
    c = np.asarray(ctr, np.uint32).copy()
    k = np.asarray(key, np.uint32).copy()
    o = np.zeros(4, np.uint32)
    lib().orc_philox4x32_10(_p(c), _p(k), _p(o))
    return o


def gbm_normals(n_pairs, n_steps, seed, stream=0, pair_offset=0):
    Z = np.empty((n_steps, n_pairs), np.float32)
    lib().orc_gbm_normals_f32(_p(Z), n_pairs, n_pairs, n_steps, seed, stream, pair_offset)
    return Z


def gbm_paths(n_paths, n_steps, S0, r, sigma, T, seed, stream=0, pair_offset=0, antithetic=1):
    S = np.empty((n_steps + 1, n_paths), np.float32)
    lib().orc_gbm_paths_f32(_p(S), n_paths, n_paths, n_steps, S0, r, sigma, T, seed, stream,
                            pair_offset, antithetic)
    return S


def gbm_paths_from_normals(z_half, S0, r, sigma, T, antithetic=1):
    z = np.ascontiguousarray(z_half, np.float32)
    N, P = z.shape
    M = 2 * P if antithetic else P
    S = np.empty((N + 1, M), np.float32)
    lib().orc_gbm_paths_from_normals_f32(_p(S), M, M, N, S0, r, sigma, T, _p(z), P, antithetic)
    return S


def heston_paths(n_paths, n_steps, S0, r, T, v0, kappa, theta, xi, rho, seed, stream=0,
                 pair_offset=0, scheme=0):
    S = np.empty((n_steps + 1, n_paths), np.float32)
    lib().orc_heston_paths_f32(_p(S), n_paths, n_paths, n_steps, S0, r, T, v0, kappa, theta, xi,
                               rho, seed, stream, pair_offset, scheme)
    return S


def heston_paths_from_normals(z1, z2, S0, r, T, v0, kappa, theta, xi, rho, scheme=0):
    z1 = np.ascontiguousarray(z1, np.float32)
    z2 = np.ascontiguousarray(z2, np.float32)
    N, P = z1.shape
    S = np.empty((N + 1, 2 * P), np.float32)
    lib().orc_heston_paths_from_normals_f32(_p(S), 2 * P, 2 * P, N, S0, r, T, v0, kappa, theta,
                                            xi, rho, _p(z1), _p(z2), P, scheme)
    return S


SEMANTICS = {"reference": 0, "textbook": 1, "two_pass": 2}


def lsm_poly(S, K, r, T, is_put, semantics="reference"):
    S = np.ascontiguousarray(S, np.float32)
    N, M = S.shape[0] - 1, S.shape[1]
    res = LsmResult()
    betas = np.zeros((N + 1, 3))
    nitm = np.zeros(N + 1, np.int64)
    sx = np.zeros(M, np.float32)
    tex = np.zeros(M, np.int32)
    rc = lib().orc_lsm_poly(_p(S), M, M, N, K, r, T, int(is_put), SEMANTICS[semantics],
                            C.byref(res), _p(betas), _p(nitm), _p(sx), _p(tex))
    assert rc == 0
    return dict(price=res.price, sum=res.sum, sumsq=res.sumsq, n_exercised=res.n_exercised,
                n_zero=res.n_zero, sum_nitm=res.sum_nitm, betas=betas, nitm=nitm, sx=sx, tex=tex)


def fold_constants(S0, K, r, sigma, T, n_steps):
    """(c0, g) of the folded storage: cK[t] = c0 g^t = S0_f32^2 exp2(2 a_f32 t) / K (orc_fold_constants)."""
    c0, g = C.c_double(), C.c_double()
    lib().orc_fold_constants(S0, K, r, sigma, T, n_steps, C.byref(c0), C.byref(g))
    return c0.value, g.value


def fold_table(n_steps, c0, g):
    t = np.zeros(n_steps + 1)
    lib().orc_fold_table(_p(t), n_steps, c0, g)
    return t


def lsm_two_pass_folded(S_half, K, r, T, is_put, c0, g):
    """Two-pass flow over BOTH partners of every antithetic pair from the first partners' paths alone
    (S_half: [N+1][n_pairs] float32; orc_lsm_two_pass_folded)."""
    S = np.ascontiguousarray(S_half, np.float32)
    N, P = S.shape[0] - 1, S.shape[1]
    res = LsmResult()
    betas = np.zeros((N + 1, 3))
    nitm = np.zeros(N + 1, np.int64)
    texa = np.zeros(P, np.int32)
    texb = np.zeros(P, np.int32)
    rc = lib().orc_lsm_two_pass_folded(_p(S), P, P, N, K, r, T, int(is_put), c0, g, C.byref(res), _p(betas), _p(nitm),
                                       _p(texa), _p(texb))
    assert rc == 0
    return dict(price=res.price, sum=res.sum, sumsq=res.sumsq, n_paths=res.n_paths, n_exercised=res.n_exercised,
                n_zero=res.n_zero, sum_nitm=res.sum_nitm, betas=betas, nitm=nitm, texa=texa, texb=texb)


def lsm_apply_frozen(S, K, r, T, is_put, betas, nitm=None):
    S = np.ascontiguousarray(S, np.float32)
    N, M = S.shape[0] - 1, S.shape[1]
    res = LsmResult()
    betas = np.ascontiguousarray(betas, np.float64)
    sx = np.zeros(M, np.float32)
    tex = np.zeros(M, np.int32)
    ni = None if nitm is None else np.ascontiguousarray(nitm, np.int64)
    lib().orc_lsm_apply_frozen(_p(S), M, M, N, K, r, T, int(is_put), _p(betas),
                               None if ni is None else _p(ni), C.byref(res), _p(sx), _p(tex))
    return dict(price=res.price, sum=res.sum, sumsq=res.sumsq, n_exercised=res.n_exercised,
                n_zero=res.n_zero, sx=sx, tex=tex)


def european_from_paths(S, K, r, T, is_put):
    S = np.ascontiguousarray(S, np.float32)
    N, M = S.shape[0] - 1, S.shape[1]
    s, q = C.c_double(), C.c_double()
    lib().orc_european_from_paths(_p(S), M, M, N, K, r, T, int(is_put), C.byref(s), C.byref(q))
    return s.value, q.value


def lsm_pass1_moments(S, K, r, T, is_put):
    S = np.ascontiguousarray(S, np.float32)
    N, M = S.shape[0] - 1, S.shape[1]
    m = np.zeros((N + 1, 8))
    lib().orc_lsm_pass1_moments(_p(S), M, M, N, K, r, T, int(is_put), _p(m))
    return m


def solve_poly2(moments):
    """moments [N+1][8] -> betas4 [N+1][4] (b0,b1,b2,n)"""
    m = np.ascontiguousarray(moments, np.float64)
    out = np.zeros((m.shape[0], 4))
    b = np.zeros(3)
    for t in range(m.shape[0]):
        row = m[t].copy()
        lib().orc_solve_poly2(_p(row), _p(b))
        out[t, :3] = b
        out[t, 3] = row[0]
    return out


def heston_terminal(n_paths, n_steps, S0, r, T, v0, kappa, theta, xi, rho, seed, stream=0,
                    pair_offset=0, scheme=2):
    ST = np.empty(n_paths, np.float32)
    lib().orc_heston_terminal_f32(_p(ST), n_paths, n_steps, S0, r, T, v0, kappa, theta, xi, rho, seed,
                                  stream, pair_offset, scheme)
    return ST
