"""ctypes wrapper over oracle/liboracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package ppqsflhe_amd.
See oracle/mkckks_oracle.h for what is restated and how parity is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, os.environ.get("ORACLE_LIB", "liboracle.so"))  # ORACLE_LIB: the sanitizer build


_FAST_PATH = os.path.join(_HERE, "libcpufast.so")  # cpu_fast.c: the restatement + a faster ReEncrypt (bench's CPU leg)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("mkckks_oracle.c", "cpu_fast.c")]
    stale = any(not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(x) for x in srcs)
                for o in (_LIB_PATH, _FAST_PATH))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_libs = {}


def lib(path=None):
    path = path or _LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        u32, u64, dbl, vp = C.c_uint32, C.c_uint64, C.c_double, C.c_void_p
        L.orc_ctx_new.restype = vp
        L.orc_ctx_new.argtypes = [u32] * 7
        L.orc_ctx_free.argtypes = [vp]
        for f in ("orc_ring_dim", "orc_num_q", "orc_num_p", "orc_alpha", "orc_beta"):
            getattr(L, f).restype = u32
            getattr(L, f).argtypes = [vp]
        L.orc_moduli.argtypes = [vp, vp]
        L.orc_roots.argtypes = [vp, vp]
        L.orc_sf.restype = dbl
        L.orc_sf.argtypes = [vp, u32]
        L.orc_sf_big.restype = dbl
        L.orc_sf_big.argtypes = [vp, u32]
        L.orc_is_prime.restype = C.c_int
        L.orc_is_prime.argtypes = [u64]
        L.orc_min_root_of_unity.restype = u64
        L.orc_min_root_of_unity.argtypes = [u64, u64]
        L.orc_ntt_fwd.argtypes = [vp, u32, vp]
        L.orc_ntt_inv.argtypes = [vp, u32, vp]
        L.orc_eval_add.argtypes = [vp, u32, vp, vp, vp]
        L.orc_rescale.argtypes = [vp, u32, vp, vp]
        L.orc_const_factors.argtypes = [vp, u32, u32, dbl, vp]
        L.orc_mult_factors.argtypes = [vp, u32, vp, vp]
        L.orc_reencrypt.argtypes = [vp, u32, vp, vp, vp]
        L.orc_modup_digits.restype = u32
        L.orc_modup_digits.argtypes = [vp, u32, vp, vp]
        L.orc_moddown.argtypes = [vp, u32, vp, vp]
        L.orc_keygen.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_rekeygen.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.orc_encode.argtypes = [vp, vp, u32, dbl, u32, vp]
        L.orc_encrypt.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
        L.orc_decrypt_decode.argtypes = [vp, u32, vp, vp, dbl, vp]
        L.orc_decrypt_core.argtypes = [vp, u32, vp, vp, vp]
        if hasattr(L, "fcpu_reencrypt"):
            L.fcpu_reencrypt.argtypes = [vp, u32, vp, vp, vp]
            L.fcpu_ntt_fwd.argtypes = [vp, u32, vp]
            L.fcpu_ntt_inv.argtypes = [vp, u32, vp]
        _libs[path] = L
    return _libs[path]


def set_threads(n):
    """Cap the OpenMP team of the restatement (bench.py's cpu_baseline states the count it used)."""
    for L in [lib()] + [v for k, v in _libs.items() if k != _LIB_PATH]:  # each library carries its own OpenMP state
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads(int(n))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


class OracleContext:
    """CKKS context with OpenFHE's FLEXIBLEAUTOEXT/HYBRID parameter selection."""

    _lib_path = None  # the restatement; FastCpuContext points at libcpufast.so

    def __init__(self, log_n, mult_depth, scaling_bits, first_bits=60, dnum=3, aux_bits=60, extra_bits=20):
        self.L_ = lib(self._lib_path)
        self.h = self.L_.orc_ctx_new(log_n, mult_depth, scaling_bits, first_bits, dnum, aux_bits, extra_bits)
        self.N = self.L_.orc_ring_dim(self.h)
        self.L = self.L_.orc_num_q(self.h)
        self.K = self.L_.orc_num_p(self.h)
        self.D = self.L + self.K
        self.alpha = self.L_.orc_alpha(self.h)
        self.beta = self.L_.orc_beta(self.h)
        m = np.zeros(self.D, dtype=np.uint64)
        r = np.zeros(self.D, dtype=np.uint64)
        self.L_.orc_moduli(self.h, _p(m))
        self.L_.orc_roots(self.h, _p(r))
        self.moduli, self.roots = m, r

    def __del__(self):
        try:
            self.L_.orc_ctx_free(self.h)
        except Exception:
            pass

    def sf(self, level):
        return self.L_.orc_sf(self.h, level)

    def sf_big(self, level):
        return self.L_.orc_sf_big(self.h, level)

    # ---- transforms
    def ntt_fwd(self, limb, a):
        a = _u64(a).copy()
        assert a.shape == (self.N,)
        self.L_.orc_ntt_fwd(self.h, limb, _p(a))
        return a

    def ntt_inv(self, limb, a):
        a = _u64(a).copy()
        assert a.shape == (self.N,)
        self.L_.orc_ntt_inv(self.h, limb, _p(a))
        return a

    # ---- ciphertext ops; ct arrays are uint64 [2][nl][N]
    def eval_add(self, a, b):
        a, b = _u64(a), _u64(b)
        nl = a.shape[1]
        out = np.empty_like(a)
        self.L_.orc_eval_add(self.h, nl, _p(a), _p(b), _p(out))
        return out

    def rescale(self, ct):
        ct = _u64(ct)
        nl = ct.shape[1]
        out = np.empty((2, nl - 1, self.N), dtype=np.uint64)
        self.L_.orc_rescale(self.h, nl, _p(ct), _p(out))
        return out

    def const_factors(self, nl, level, operand):
        f = np.zeros(nl, dtype=np.uint64)
        self.L_.orc_const_factors(self.h, nl, level, float(operand), _p(f))
        return f

    def mult_factors(self, ct, factors):
        ct = _u64(ct).copy()
        f = _u64(factors)
        self.L_.orc_mult_factors(self.h, ct.shape[1], _p(f), _p(ct))
        return ct

    def reencrypt(self, ct, evk):
        ct, evk = _u64(ct), _u64(evk)
        nl = ct.shape[1]
        assert evk.shape == (self.beta, 2, self.D, self.N)
        out = np.empty_like(ct)
        self.L_.orc_reencrypt(self.h, nl, _p(ct), _p(evk), _p(out))
        return out

    def modup_digits(self, c1):
        c1 = _u64(c1)
        nl = c1.shape[0]
        dig = np.zeros((self.beta, nl + self.K, self.N), dtype=np.uint64)
        nparts = self.L_.orc_modup_digits(self.h, nl, _p(c1), _p(dig))
        return dig[:nparts]

    def moddown(self, x):
        x = _u64(x)
        nl = x.shape[0] - self.K
        out = np.empty((nl, self.N), dtype=np.uint64)
        self.L_.orc_moddown(self.h, nl, _p(x), _p(out))
        return out

    # ---- keys
    def keygen(self, s_tern, a_eval, e):
        s = np.ascontiguousarray(s_tern, dtype=np.int8)
        a = _u64(a_eval)
        e = np.ascontiguousarray(e, dtype=np.int32)
        pk = np.empty((2, self.D, self.N), dtype=np.uint64)
        sk = np.empty((self.D, self.N), dtype=np.uint64)
        self.L_.orc_keygen(self.h, _p(s), _p(a), _p(e), _p(pk), _p(sk))
        return pk, sk

    def rekeygen(self, s_old, pk_new, u, e0, e1):
        s = np.ascontiguousarray(s_old, dtype=np.int8)
        pk = _u64(pk_new)
        u = np.ascontiguousarray(u, dtype=np.int8)
        e0 = np.ascontiguousarray(e0, dtype=np.int32)
        e1 = np.ascontiguousarray(e1, dtype=np.int32)
        assert u.shape == (self.beta, self.N)
        evk = np.empty((self.beta, 2, self.D, self.N), dtype=np.uint64)
        self.L_.orc_rekeygen(self.h, _p(s), _p(pk), _p(u), _p(e0), _p(e1), _p(evk))
        return evk

    # ---- client endpoints
    def encode(self, vals, scale, nl):
        v = np.ascontiguousarray(vals, dtype=np.float64)
        pt = np.empty((nl, self.N), dtype=np.uint64)
        self.L_.orc_encode(self.h, _p(v), v.size, float(scale), nl, _p(pt))
        return pt

    def encrypt(self, pk, pt, v, e0, e1):
        pk, pt = _u64(pk), _u64(pt)
        nl = pt.shape[0]
        v = np.ascontiguousarray(v, dtype=np.int8)
        e0 = np.ascontiguousarray(e0, dtype=np.int32)
        e1 = np.ascontiguousarray(e1, dtype=np.int32)
        ct = np.empty((2, nl, self.N), dtype=np.uint64)
        self.L_.orc_encrypt(self.h, nl, _p(pk), _p(pt), _p(v), _p(e0), _p(e1), _p(ct))
        return ct

    def decrypt_core(self, ct, sk_eval):
        ct, sk = _u64(ct), _u64(sk_eval)
        nl = ct.shape[1]
        m = np.empty((nl, self.N), dtype=np.uint64)
        self.L_.orc_decrypt_core(self.h, nl, _p(ct), _p(sk), _p(m))
        return m

    def decrypt_decode(self, ct, sk_eval, scale):
        ct, sk = _u64(ct), _u64(sk_eval)
        nl = ct.shape[1]
        out = np.empty(self.N // 2, dtype=np.float64)
        self.L_.orc_decrypt_decode(self.h, nl, _p(ct), _p(sk), float(scale), _p(out))
        return out


# ---- seeded samplers shared by tests and bench (numpy; not part of the product)

class FastCpuContext(OracleContext):
    """The same context out of libcpufast.so (cpu_fast.c = the restatement + fcpu_reencrypt): ReEncrypt the way OpenFHE's
    native backend carries it out (lazy butterflies, Shoup/Barrett constants, tables built once).  bench.py's cpu_baseline
    times this one; tests compare it with OracleContext.reencrypt word for word."""
    _lib_path = _FAST_PATH

    def reencrypt(self, ct, evk):
        ct, evk = _u64(ct), _u64(evk)
        nl = ct.shape[1]
        assert evk.shape == (self.beta, 2, self.D, self.N)
        out = np.empty_like(ct)
        self.L_.fcpu_reencrypt(self.h, nl, _p(ct), _p(evk), _p(out))
        return out

    def ntt_fwd(self, limb, a):
        a = _u64(a).copy()
        self.L_.fcpu_ntt_fwd(self.h, limb, _p(a))
        return a

    def ntt_inv(self, limb, a):
        a = _u64(a).copy()
        self.L_.fcpu_ntt_inv(self.h, limb, _p(a))
        return a


def sample_ternary(rng, n):
    """[upstream] TernaryUniformGeneratorImpl: uniform over {-1,0,1}."""
    return rng.integers(-1, 2, size=n, dtype=np.int8)


def sample_gauss(rng, n, sigma=3.19):
    """[upstream] DiscreteGaussianGeneratorImpl with sigma = 3.19 (CC.json 'dp')."""
    return np.rint(rng.normal(0.0, sigma, size=n)).astype(np.int32)


def sample_uniform(rng, moduli, n):
    """[upstream] DiscreteUniformGeneratorImpl: one uniform residue vector per limb."""
    out = np.empty((len(moduli), n), dtype=np.uint64)
    for i, q in enumerate(moduli):
        out[i] = rng.integers(0, int(q), size=n, dtype=np.uint64)
    return out
