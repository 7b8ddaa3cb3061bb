"""ctypes binding of libstarkhip.so (the C ABI declared in include/starkhip.h).

There is no CPU fallback: if the shared library is missing or a call fails, an exception is
raised.  ``import torch`` happens first when torch is installed so that this process ends up
with a single HIP runtime (torch bundles its own libamdhip64 under the same soname).
"""
from __future__ import annotations

import ctypes
import os

try:  # plumbing only: device tensors and torch.distributed for the benchmarks
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for the C ABI itself
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
# STARKHIP_LIB selects another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("STARKHIP_LIB") or os.path.join(_HERE, "libstarkhip.so")

STARK_OK = 0
STARK_ERR_INVALID = -1
STARK_ERR_HIP = -2
STARK_ERR_NOMEM = -3
STARK_ERR_UNSUPPORTED = -4

EXPORTS = [
    "stark_version", "stark_ctx_create", "stark_ctx_destroy", "stark_last_error", "stark_ctx_stream",
    "stark_ctx_synchronize", "stark_fft_inv", "stark_fft", "stark_ntt_batch", "stark_ntt_batch_dev",
    "stark_merkle_verify_batch", "stark_merkle_verify_batch_dev", "stark_field_op",
    "stark_fill_random_field_dev", "stark_twiddle_mul_dev", "stark_permute3_dev",
    "stark_mimc_verify_batch", "stark_pseudorandom_indices", "stark_pseudorandom_indices_dev",
    "stark_merkle_build", "stark_merkle_build_dev",
    "stark_mimc_proof_size", "stark_mimc_prove_batch", "stark_mimc_prove_batch_dev", "stark_mimc_verify_batch_io", "stark_mimc_verify_batch_dev", "stark_mimc_batch", "stark_mimc_batch_dev",
    "stark_mimc_verify_batch_diag", "stark_ntt_batch_dev_static",
    "stark_ntt_sharded_stage_a_dev", "stark_ntt_sharded_stage_c_dev", "stark_ntt_sharded_dev", "stark_ntt_columns_dev",
    "stark_be32_to_limbs", "stark_limbs_to_be32",
]


# int (*stark_all_to_all_fn)(void* user, const void* d_send, void* d_recv, size_t block_bytes, void* stream)
ALL_TO_ALL_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                 ctypes.c_void_p)


class StarkError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libstarkhip error {code}: {message}")
        self.code = code


class InvalidArgument(StarkError, ValueError):
    """The call was rejected (shapes on which the reference implementation panics)."""


_lib = None
_other_libs = {}


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = load_from(LIB_PATH)
    return _lib


def load_from(path: str) -> ctypes.CDLL:
    """Another build of the library under its own handle (the tests' libstarkhip_diag.so); pass it to Context(lib=...)."""
    path = os.path.abspath(path)
    if path not in _other_libs:
        _other_libs[path] = _bind(path)
    return _other_libs[path]


def _bind(path: str) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `make -C stark_verifier_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback")
    L = ctypes.CDLL(path)
    vp, sz, u32, u64, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
    L.stark_version.restype = i32
    L.stark_ctx_create.argtypes = [i32, ctypes.POINTER(vp)]
    L.stark_ctx_create.restype = i32
    L.stark_ctx_destroy.argtypes = [vp]
    L.stark_ctx_destroy.restype = None
    L.stark_last_error.argtypes = [vp]
    L.stark_last_error.restype = ctypes.c_char_p
    L.stark_ctx_stream.argtypes = [vp]
    L.stark_ctx_stream.restype = vp
    L.stark_ctx_synchronize.argtypes = [vp]
    L.stark_ctx_synchronize.restype = i32
    L.stark_fft_inv.argtypes = [vp, vp, sz, vp, vp, vp]
    L.stark_fft_inv.restype = i32
    L.stark_fft.argtypes = [vp, vp, sz, vp, vp, vp]
    L.stark_fft.restype = i32
    L.stark_ntt_batch.argtypes = [vp, vp, ctypes.c_uint, sz, vp, i32]
    L.stark_ntt_batch.restype = i32
    L.stark_ntt_batch_dev.argtypes = [vp, vp, vp, ctypes.c_uint, sz, vp, i32, vp]
    L.stark_ntt_batch_dev.restype = i32
    L.stark_ntt_batch_dev_static.argtypes = [vp, vp, vp, ctypes.c_uint, sz, vp, i32, vp]
    L.stark_ntt_batch_dev_static.restype = i32
    L.stark_ntt_sharded_stage_a_dev.argtypes = [vp, vp, vp, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, vp, i32, vp]
    L.stark_ntt_sharded_stage_a_dev.restype = i32
    L.stark_ntt_sharded_stage_c_dev.argtypes = [vp, vp, vp, ctypes.c_uint, ctypes.c_uint, vp, i32, vp]
    L.stark_ntt_sharded_stage_c_dev.restype = i32
    L.stark_ntt_sharded_dev.argtypes = [vp, vp, vp, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, vp, i32, ALL_TO_ALL_FN, vp, vp]
    L.stark_ntt_sharded_dev.restype = i32
    L.stark_ntt_columns_dev.argtypes = [vp, vp, vp, ctypes.c_uint, sz, vp, i32, vp]
    L.stark_ntt_columns_dev.restype = i32
    L.stark_be32_to_limbs.argtypes = [vp, sz, vp]
    L.stark_be32_to_limbs.restype = None
    L.stark_limbs_to_be32.argtypes = [vp, sz, vp]
    L.stark_limbs_to_be32.restype = None
    L.stark_merkle_verify_batch.argtypes = [vp, vp, vp, u32, vp, u32, vp, vp, u32, vp, sz, vp, vp]
    L.stark_merkle_verify_batch.restype = i32
    L.stark_merkle_verify_batch_dev.argtypes = [vp, vp, vp, u32, vp, u32, vp, vp, vp, sz, vp, vp, vp]
    L.stark_merkle_verify_batch_dev.restype = i32
    L.stark_field_op.argtypes = [vp, i32, vp, vp, vp, sz]
    L.stark_field_op.restype = i32
    L.stark_fill_random_field_dev.argtypes = [vp, vp, sz, u64, vp]
    L.stark_fill_random_field_dev.restype = i32
    L.stark_twiddle_mul_dev.argtypes = [vp, vp, sz, sz, u64, u64, ctypes.c_uint, vp, i32, vp]
    L.stark_twiddle_mul_dev.restype = i32
    L.stark_permute3_dev.argtypes = [vp, vp, vp, sz, sz, sz, vp]
    L.stark_permute3_dev.restype = i32
    L.stark_mimc_verify_batch.argtypes = [vp, vp, vp, sz, vp, u64, vp, sz, vp, vp, vp]
    L.stark_mimc_verify_batch.restype = i32
    L.stark_mimc_verify_batch_io.argtypes = [vp, vp, vp, sz, vp, u64, vp, sz, vp, vp, vp]
    L.stark_mimc_verify_batch_io.restype = i32
    L.stark_mimc_verify_batch_diag.argtypes = [vp, vp, vp, sz, vp, u64, vp, sz, vp, ctypes.c_uint, vp, vp]
    L.stark_mimc_verify_batch_diag.restype = i32
    L.stark_mimc_verify_batch_dev.argtypes = [vp, vp, u64, sz, vp, u64, vp, sz, vp, vp, vp]
    L.stark_mimc_verify_batch_dev.restype = i32
    L.stark_pseudorandom_indices.argtypes = [vp, vp, sz, u32, u32, u32, vp]
    L.stark_pseudorandom_indices.restype = i32
    L.stark_pseudorandom_indices_dev.argtypes = [vp, vp, sz, u32, u32, u32, vp, vp]
    L.stark_pseudorandom_indices_dev.restype = i32
    L.stark_merkle_build.argtypes = [vp, vp, ctypes.c_uint, vp]
    L.stark_merkle_build.restype = i32
    L.stark_merkle_build_dev.argtypes = [vp, vp, ctypes.c_uint, vp]
    L.stark_merkle_build_dev.restype = i32
    L.stark_mimc_batch.argtypes = [vp, vp, sz, u64, vp, sz, vp]
    L.stark_mimc_batch.restype = i32
    L.stark_mimc_batch_dev.argtypes = [vp, vp, sz, u64, vp, sz, vp, vp]
    L.stark_mimc_batch_dev.restype = i32
    L.stark_mimc_proof_size.argtypes = [u64, sz]
    L.stark_mimc_proof_size.restype = sz
    L.stark_mimc_prove_batch.argtypes = [vp, vp, sz, u64, vp, sz, vp, vp]
    L.stark_mimc_prove_batch.restype = i32
    L.stark_mimc_prove_batch_dev.argtypes = [vp, vp, sz, u64, vp, sz, vp, u64, vp, vp]
    L.stark_mimc_prove_batch_dev.restype = i32
    return L


class Context:
    """Owns one ``stark_ctx`` (device buffers, stream, cached twiddle tables)."""

    def __init__(self, device: int = 0, lib: ctypes.CDLL | None = None):
        self._lib = lib or load()
        h = ctypes.c_void_p()
        rc = self._lib.stark_ctx_create(device, ctypes.byref(h))
        if rc != STARK_OK:
            raise StarkError(rc, f"stark_ctx_create(device={device}) failed (is a HIP device visible?)")
        self._h = h
        self.device = device

    def check(self, rc: int):
        if rc == STARK_OK:
            return
        msg = self._lib.stark_last_error(self._h).decode("utf-8", "replace")
        if rc == STARK_ERR_INVALID:
            raise InvalidArgument(rc, msg)
        raise StarkError(rc, msg)

    @property
    def handle(self):
        return self._h

    @property
    def lib(self):
        return self._lib

    @property
    def stream(self) -> int:
        return self._lib.stark_ctx_stream(self._h) or 0

    def synchronize(self):
        self.check(self._lib.stark_ctx_synchronize(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.stark_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def current_device() -> int:
    """The GPU this process works on. If the caller has touched torch.cuda already (torch.cuda.set_device, a tensor
    on a device, anything that initialises it), that choice stands: torch's current device. Only a process that has
    not, started by torchrun with every GPU visible, takes its LOCAL_RANK (when the node shows that many GPUs), so that
    the ranks of a one-process-per-GPU job do not all land on GPU 0 by default. Callers that keep their tensors or
    their communicator on another device than these rules give pass `device` explicitly."""
    lr = os.environ.get("LOCAL_RANK")
    if torch is not None and torch.cuda.is_available():
        if torch.cuda.is_initialized():
            return torch.cuda.current_device()
        if lr is not None and lr.isdigit() and int(lr) < torch.cuda.device_count():
            return int(lr)
        return torch.cuda.current_device()
    return 0


def default_context(device: int | None = None) -> Context:
    """The process-wide context of `device` (None: current_device())."""
    if device is None:
        device = current_device()
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = _default_ctx[device] = Context(device)
    return ctx
