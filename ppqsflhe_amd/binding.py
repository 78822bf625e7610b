"""ctypes binding of libmkckks_hip.so (C-ABI: include/mkckks.h).

Mirrors, for the hot path only, the OpenFHE CryptoContext calls the reference's
C++ mains make:
    cc->ReEncrypt(ct, reKey)          -> Context.reencrypt          changeCipherDomain.cpp:74
    cc->EvalAdd(ct1, ct2)             -> Context.eval_add           aggregateEncryptedWeights.cpp:82
    cc->EvalMult(ct, 0.5)             -> Context.rescale_mult_const aggregateEncryptedWeights.cpp:83
    cc->KeyGen() / ReKeyGen / Encrypt / Decrypt -> keygen / rekeygen / encrypt / decrypt
Buffers are DeviceBuffer objects (HBM, limb-major uint64) or anything with a
``data_ptr()`` (a torch tensor on the GPU).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("MKCKKS_LIB") or os.path.join(_HERE, "libmkckks_hip.so")  # override: A/B builds


class MkckksError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mkckks error {code}: {msg}")
        self.code = code


class _Params(C.Structure):
    _fields_ = [("log_n", C.c_uint32), ("mult_depth", C.c_uint32), ("scaling_bits", C.c_uint32),
                ("first_bits", C.c_uint32), ("dnum", C.c_uint32), ("aux_bits", C.c_uint32),
                ("extra_bits", C.c_uint32), ("device", C.c_int32)]


class _Info(C.Structure):
    _fields_ = [("ring_dim", C.c_uint32), ("num_q", C.c_uint32), ("num_p", C.c_uint32),
                ("alpha", C.c_uint32), ("beta", C.c_uint32), ("slots", C.c_uint32)]


# every symbol include/mkckks.h declares: name -> (restype, argtypes)
_vp, _u32, _u64p, _sz, _int, _dbl = C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_int, C.c_double
SYMBOLS = {
    "mkckks_last_error": (C.c_char_p, []),
    "mkckks_version": (C.c_char_p, []),
    "mkckks_ctx_create": (_int, [C.POINTER(_Params), C.POINTER(_vp)]),
    "mkckks_ctx_destroy": (_int, [_vp]),
    "mkckks_ctx_info": (_int, [_vp, C.POINTER(_Info)]),
    "mkckks_ctx_moduli": (_int, [_vp, _u64p]),
    "mkckks_ctx_roots": (_int, [_vp, _u64p]),
    "mkckks_ctx_arith": (_int, [_vp, _vp]),
    "mkckks_scaling_factor": (_int, [_vp, _u32, _int, C.POINTER(_dbl)]),
    "mkckks_set_stream": (_int, [_vp, _vp]),
    "mkckks_sync": (_int, [_vp]),
    "mkckks_dev_alloc": (_int, [_vp, _sz, C.POINTER(_vp)]),
    "mkckks_dev_free": (_int, [_vp, _vp]),
    "mkckks_upload": (_int, [_vp, _vp, _vp, _sz]),
    "mkckks_download": (_int, [_vp, _vp, _vp, _sz]),
    "mkckks_host_alloc": (_int, [_vp, _sz, C.POINTER(_vp)]),
    "mkckks_host_free": (_int, [_vp, _vp]),
    "mkckks_upload_async": (_int, [_vp, _vp, _vp, _sz, C.POINTER(C.c_uint64)]),
    "mkckks_download_async": (_int, [_vp, _vp, _vp, _sz, C.POINTER(C.c_uint64)]),
    "mkckks_copy_done": (_int, [_vp, C.c_uint64, C.POINTER(_int)]),
    "mkckks_copy_wait": (_int, [_vp, C.c_uint64]),
    "mkckks_fence_uploads": (_int, [_vp]),
    "mkckks_fence_compute": (_int, [_vp]),
    "mkckks_count_noncanonical": (_int, [_vp, _vp, _u32, _u32, C.POINTER(C.c_uint64)]),
    "mkckks_debug_stamps": (_int, [_vp, _vp, _u32, C.POINTER(_sz)]),
    "mkckks_ntt_forward_batch": (_int, [_vp, _vp, _u32, _u32, _int]),
    "mkckks_ntt_inverse_batch": (_int, [_vp, _vp, _u32, _u32, _int]),
    "mkckks_eval_add_batch": (_int, [_vp, _vp, _vp, _vp, _u32, _u32]),
    "mkckks_eval_sum_batch": (_int, [_vp, _vp, _vp, _u32, _u32, _u32]),
    "mkckks_rescale_mult_const_batch": (_int, [_vp, _vp, _vp, _u32, _u32, _dbl]),
    "mkckks_rescale_batch": (_int, [_vp, _vp, _vp, _u32, _u32]),
    "mkckks_mult_const_batch": (_int, [_vp, _vp, _u32, _u32, _dbl]),
    "mkckks_reencrypt_batch": (_int, [_vp, _vp, _vp, _vp, _u32, _u32]),
    "mkckks_reencrypt_accumulate_batch": (_int, [_vp, _vp, _vp, _vp, _u32, _u32]),
    "mkckks_reencrypt_sum_batch": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32]),
    "mkckks_modup_batch": (_int, [_vp, _vp, _vp, _u32, _u32]),
    "mkckks_moddown_batch": (_int, [_vp, _vp, _vp, _u32, _u32]),
    "mkckks_keygen": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "mkckks_rekeygen": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mkckks_encrypt_batch": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32]),
    "mkckks_lift_ntt_batch": (_int, [_vp, _vp, _vp, _u32, _u32]),
    "mkckks_decrypt_batch": (_int, [_vp, _vp, _vp, _vp, _u32, _u32]),
    "mkckks_sample_ternary": (_int, [_vp, _vp, _sz, C.c_char_p, _u32]),
    "mkckks_sample_gauss": (_int, [_vp, _vp, _sz, _dbl, C.c_char_p, _u32]),
    "mkckks_sample_uniform": (_int, [_vp, _vp, _u32, _u32, _int, C.c_char_p, _u32]),
    "mkckks_chacha20_block": (_int, [_vp, _vp, C.c_char_p, _u32, C.POINTER(_u32)]),
    "mkckks_encode_batch": (_int, [_vp, _vp, _vp, _u32, _u32, _dbl]),
    "mkckks_decode_batch": (_int, [_vp, _vp, _vp, _u32, _u32, _dbl]),
    "mkckks_reduce_mod_batch": (_int, [_vp, _vp, _u32, _u32, _u32]),
    "mkckks_comm_unique_id": (_int, [_vp]),
    "mkckks_comm_create": (_int, [_vp, _vp, _int, _int, C.POINTER(_vp)]),
    "mkckks_comm_destroy": (_int, [_vp, _vp]),
    "mkckks_reduce_scatter_sum_mod": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32]),
    "mkckks_comm_library": (C.c_char_p, []),
    "mkckks_ctx_twiddles": (_int, [_vp, _u32, _int, _u64p]),
}
COMM_ID_BYTES = 128

_lib = None


def lib_path():
    return _LIB


def hip_runtimes_loaded():
    """Distinct libamdhip64 files mapped into this process (more than one is the "no HIP device" trap)."""
    paths = []
    try:
        for line in open("/proc/self/maps"):
            if "libamdhip64" in line:
                p = line[line.index("/"):].strip()
                if p not in paths:
                    paths.append(p)
    except (OSError, ValueError):
        pass
    return paths


def load_library():
    """Load the in-tree HIP library; raise (never fall back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise ImportError(
                f"{_LIB} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C ppqsflhe_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch-ROCm ships its own libamdhip64: when both end up in one process the FIRST HIP runtime loaded owns
        # the device and the other reports "no ROCm-capable device".  Tensors from torch are this binding's device
        # buffers (bench.py, multi-GPU), so let torch load first whenever it is installed.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass  # no torch in this environment: the library's own runtime is the only one
        L = C.CDLL(_LIB)
        dup = hip_runtimes_loaded()
        if len(dup) > 1:
            import warnings
            warnings.warn("two HIP runtimes in this process (" + ", ".join(dup) + "): the one initialised second sees no "
                          "device; import torch before ppqsflhe_amd, or do not mix ROCm installations", RuntimeWarning)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def sampler_key(key):
    if isinstance(key, int):
        key = int(key).to_bytes(32, "little")
    key = bytes(key)
    if len(key) != 32:
        raise ValueError("sampler key must be 32 bytes")
    return key


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):  # torch tensor on the device
        return x.data_ptr()
    if isinstance(x, int):
        return x
    raise TypeError(f"expected a device buffer, got {type(x)}")


class DeviceBuffer:
    """A typed, shaped view of HBM owned by a Context."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        ctx._check(ctx._L.mkckks_dev_alloc(ctx._h, self.nbytes, C.byref(p)))
        self.ptr = p.value
        self._owned = True

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.nbytes == self.nbytes, (arr.shape, self.shape)
        self.ctx._check(self.ctx._L.mkckks_upload(self.ctx._h, self.ptr, arr.ctypes.data, self.nbytes))
        return self

    def to_host(self):
        out = np.empty(self.shape, dtype=self.dtype)
        self.ctx._check(self.ctx._L.mkckks_download(self.ctx._h, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def view(self, offset_elems, shape):
        """A non-owning sub-view (element offset) of this buffer."""
        v = object.__new__(DeviceBuffer)
        v.ctx, v.shape, v.dtype = self.ctx, tuple(shape), self.dtype
        v.nbytes = int(np.prod(v.shape, dtype=np.int64)) * v.dtype.itemsize
        v.ptr = self.ptr + offset_elems * self.dtype.itemsize
        v._owned = False
        v._parent = self
        return v

    def free(self):
        if getattr(self, "_owned", False) and self.ptr and self.ctx._h:
            self.ctx._L.mkckks_dev_free(self.ctx._h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """Device-resident CKKS context (moduli, twiddles, CRT tables in HBM)."""

    def __init__(self, log_n, mult_depth, scaling_bits, first_bits=60, dnum=3, aux_bits=60, extra_bits=20,
                 device=0):
        self._L = load_library()
        self._h = None
        p = _Params(log_n, mult_depth, scaling_bits, first_bits, dnum, aux_bits, extra_bits, device)
        h = C.c_void_p()
        self._check(self._L.mkckks_ctx_create(C.byref(p), C.byref(h)))
        self._h = h.value
        info = _Info()
        self._check(self._L.mkckks_ctx_info(self._h, C.byref(info)))
        self.N, self.L, self.K = info.ring_dim, info.num_q, info.num_p
        self.D = self.L + self.K
        self.alpha, self.beta, self.slots = info.alpha, info.beta, info.slots
        self.device = device
        self.moduli = np.zeros(self.D, dtype=np.uint64)
        self.roots = np.zeros(self.D, dtype=np.uint64)
        self._check(self._L.mkckks_ctx_moduli(self._h, self.moduli.ctypes.data))
        self._check(self._L.mkckks_ctx_roots(self._h, self.roots.ctypes.data))
        # arithmetic class of every limb on the device: 0 integer (Shoup), 1 fp64, 2 integer (pseudo-Mersenne)
        self.arith = np.zeros(self.D, dtype=np.uint8)
        self._check(self._L.mkckks_ctx_arith(self._h, self.arith.ctypes.data))

    def _check(self, rc):
        if rc != 0:
            raise MkckksError(rc, self._L.mkckks_last_error().decode())

    def close(self):
        if self._h:
            for p in list(getattr(self, "_pinned", {}).values()):  # pinned buffers the caller did not give back
                self._L.mkckks_host_free(self._h, p)
            self._pinned = {}
            self._L.mkckks_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters
    def sf(self, level):
        out = C.c_double()
        self._check(self._L.mkckks_scaling_factor(self._h, level, 0, C.byref(out)))
        return out.value

    def sf_big(self, level):
        out = C.c_double()
        self._check(self._L.mkckks_scaling_factor(self._h, level, 1, C.byref(out)))
        return out.value

    def twiddles(self, limb, inverse=False):
        out = np.zeros(self.N, dtype=np.uint64)
        self._check(self._L.mkckks_ctx_twiddles(self._h, limb, int(inverse), out.ctypes.data))
        return out

    def num_parts(self, nl):
        return min(self.beta, -(-nl // self.alpha))

    # ---- memory / stream
    def empty(self, shape, dtype=np.uint64):
        return DeviceBuffer(self, shape, dtype)

    def to_device(self, arr, dtype=None):
        arr = np.ascontiguousarray(arr, dtype=dtype or arr.dtype)
        return DeviceBuffer(self, arr.shape, arr.dtype).upload(arr)

    def set_stream(self, stream_handle):
        self._check(self._L.mkckks_set_stream(self._h, stream_handle))

    def sync(self):
        self._check(self._L.mkckks_sync(self._h))

    # ---- transforms (in place)
    def ntt_forward(self, d_polys, n_polys, nl, with_p=False):
        self._check(self._L.mkckks_ntt_forward_batch(self._h, _ptr(d_polys), n_polys, nl, int(with_p)))

    def ntt_inverse(self, d_polys, n_polys, nl, with_p=False):
        self._check(self._L.mkckks_ntt_inverse_batch(self._h, _ptr(d_polys), n_polys, nl, int(with_p)))

    # ---- aggregation
    def eval_add(self, a, b, out, n_ct, nl):
        self._check(self._L.mkckks_eval_add_batch(self._h, _ptr(a), _ptr(b), _ptr(out), n_ct, nl))

    def eval_sum(self, inp, out, n_clients, n_ct, nl):
        self._check(self._L.mkckks_eval_sum_batch(self._h, _ptr(inp), _ptr(out), n_clients, n_ct, nl))

    def rescale_mult_const(self, inp, out, n_ct, nl, operand):
        self._check(self._L.mkckks_rescale_mult_const_batch(self._h, _ptr(inp), _ptr(out), n_ct, nl, float(operand)))

    def rescale(self, inp, out, n_ct, nl):
        self._check(self._L.mkckks_rescale_batch(self._h, _ptr(inp), _ptr(out), n_ct, nl))

    def mult_const(self, ct, n_ct, nl, operand):
        self._check(self._L.mkckks_mult_const_batch(self._h, _ptr(ct), n_ct, nl, float(operand)))

    def reduce_mod(self, ct, n_ct, nl, n_terms):
        self._check(self._L.mkckks_reduce_mod_batch(self._h, _ptr(ct), n_ct, nl, n_terms))

    # ---- I/O pipeline (pinned host buffers, asynchronous copies with tickets, fences; include/mkckks.h)
    def host_alloc(self, nbytes):
        """Pinned host buffer as a numpy uint8 array (free it with host_free(array))."""
        p = C.c_void_p()
        self._check(self._L.mkckks_host_alloc(self._h, nbytes, C.byref(p)))
        arr = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p.value))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        self._check(self._L.mkckks_host_free(self._h, self._pinned.pop(arr.ctypes.data)))

    def upload_async(self, dev, host_arr, nbytes=None):
        t = C.c_uint64()
        self._check(self._L.mkckks_upload_async(self._h, _ptr(dev), host_arr.ctypes.data,
                                                host_arr.nbytes if nbytes is None else nbytes, C.byref(t)))
        return t.value

    def download_async(self, host_arr, dev, nbytes=None):
        t = C.c_uint64()
        self._check(self._L.mkckks_download_async(self._h, host_arr.ctypes.data, _ptr(dev),
                                                  host_arr.nbytes if nbytes is None else nbytes, C.byref(t)))
        return t.value

    def copy_done(self, ticket):
        d = C.c_int()
        self._check(self._L.mkckks_copy_done(self._h, ticket, C.byref(d)))
        return bool(d.value)

    def copy_wait(self, ticket):
        self._check(self._L.mkckks_copy_wait(self._h, ticket))

    def fence_uploads(self):
        self._check(self._L.mkckks_fence_uploads(self._h))

    def fence_compute(self):
        self._check(self._L.mkckks_fence_compute(self._h))

    def count_noncanonical(self, ct, n_ct, nl):
        c = C.c_uint64()
        self._check(self._L.mkckks_count_noncanonical(self._h, _ptr(ct), n_ct, nl, C.byref(c)))
        return c.value

    # ---- RCCL exchange of the per-GPU partial sums (behind the C-ABI; librccl resolved by the library)
    def comm_unique_id(self):
        """Rank 0: the MKCKKS_COMM_ID_BYTES bootstrap bytes (ncclGetUniqueId) to hand to every other rank."""
        buf = C.create_string_buffer(COMM_ID_BYTES)
        self._check(self._L.mkckks_comm_unique_id(buf))
        return buf.raw

    def comm_create(self, unique_id, n_ranks, rank):
        """Collective: an ncclComm_t (opaque handle) for this context's device."""
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("communicator id must be MKCKKS_COMM_ID_BYTES long")
        h = C.c_void_p()
        self._check(self._L.mkckks_comm_create(self._h, C.c_char_p(unique_id), n_ranks, rank, C.byref(h)))
        return h.value

    def comm_destroy(self, comm):
        self._check(self._L.mkckks_comm_destroy(self._h, comm))

    def reduce_scatter_sum_mod(self, comm, partial, shard, n_ct_shard, nl, n_ranks):
        self._check(self._L.mkckks_reduce_scatter_sum_mod(self._h, comm, _ptr(partial), _ptr(shard), n_ct_shard, nl,
                                                          n_ranks))

    def comm_library(self):
        return self._L.mkckks_comm_library().decode()

    # ---- proxy re-encryption
    def reencrypt(self, ct, evk, out, n_ct, nl):
        self._check(self._L.mkckks_reencrypt_batch(self._h, _ptr(ct), _ptr(evk), _ptr(out), n_ct, nl))

    def reencrypt_accumulate(self, ct, evk, acc, n_ct, nl):
        self._check(self._L.mkckks_reencrypt_accumulate_batch(self._h, _ptr(ct), _ptr(evk), _ptr(acc), n_ct, nl))

    def reencrypt_sum(self, cts, evks, out, n_clients, n_ct, nl):
        self._check(self._L.mkckks_reencrypt_sum_batch(self._h, _ptr(cts), _ptr(evks), _ptr(out), n_clients, n_ct, nl))

    def modup(self, c1, digits, n, nl):
        self._check(self._L.mkckks_modup_batch(self._h, _ptr(c1), _ptr(digits), n, nl))

    def moddown(self, inp, out, n, nl):
        self._check(self._L.mkckks_moddown_batch(self._h, _ptr(inp), _ptr(out), n, nl))

    # ---- keys and client endpoints
    def keygen(self, s, a, e, pk, sk):
        self._check(self._L.mkckks_keygen(self._h, _ptr(s), _ptr(a), _ptr(e), _ptr(pk), _ptr(sk)))

    def rekeygen(self, s_old, pk_new, u, e0, e1, evk):
        self._check(self._L.mkckks_rekeygen(self._h, _ptr(s_old), _ptr(pk_new), _ptr(u), _ptr(e0), _ptr(e1), _ptr(evk)))

    def encrypt(self, pk, pt, v, e0, e1, ct, n_ct, nl):
        self._check(self._L.mkckks_encrypt_batch(self._h, _ptr(pk), _ptr(pt), _ptr(v), _ptr(e0), _ptr(e1), _ptr(ct),
                                                 n_ct, nl))

    def lift_ntt(self, coef, out, n, nl):
        self._check(self._L.mkckks_lift_ntt_batch(self._h, _ptr(coef), _ptr(out), n, nl))

    # ---- samplers: `key` is the 32-byte ChaCha20 key (bytes); an int is accepted for tests and expanded
    # little-endian with zero padding (a test vector, NOT a way to key production randomness)
    def sample_ternary(self, out, count, key, stream_id=0):
        self._check(self._L.mkckks_sample_ternary(self._h, _ptr(out), count, sampler_key(key), stream_id))

    def sample_gauss(self, out, count, sigma, key, stream_id=0):
        self._check(self._L.mkckks_sample_gauss(self._h, _ptr(out), count, float(sigma), sampler_key(key), stream_id))

    def sample_uniform(self, out, n_polys, nl, with_p, key, stream_id=0):
        self._check(self._L.mkckks_sample_uniform(self._h, _ptr(out), n_polys, nl, int(with_p), sampler_key(key), stream_id))

    def chacha20_block(self, key, counter, nonce):
        out = self.empty((16,), np.uint32)
        n3 = (C.c_uint32 * 3)(*nonce)
        self._check(self._L.mkckks_chacha20_block(self._h, out.ptr, sampler_key(key), counter, n3))
        return out.to_host()

    def encode(self, vals, pt, n, nl, scale):
        self._check(self._L.mkckks_encode_batch(self._h, _ptr(vals), _ptr(pt), n, nl, float(scale)))

    def decode(self, m, vals, n, nl, scale):
        self._check(self._L.mkckks_decode_batch(self._h, _ptr(m), _ptr(vals), n, nl, float(scale)))

    def decrypt(self, ct, sk, m, n_ct, nl):
        self._check(self._L.mkckks_decrypt_batch(self._h, _ptr(ct), _ptr(sk), _ptr(m), n_ct, nl))
