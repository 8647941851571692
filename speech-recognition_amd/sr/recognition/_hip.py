# -*- coding: utf-8 -*-
"""ctypes binding of libgmmhmm.so (the C ABI declared in include/gmmhmm.h).

This is the only door from the Python mirror of `sr.recognition` to the GPU.
There is NO CPU fallback: if the library is missing or no MI355X is visible,
every entry point raises `BackendError`.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GMMHMM_LIB") or os.path.normpath(
    os.path.join(_HERE, "..", "..", "lib", "libgmmhmm.so"))

GH_F32, GH_F64 = 0, 1
GH_ERR_SELF_POINTER = -5
GH_ERR_UNSUPPORTED = -6
GH_ERR_COMM = -7

_c_i32p = C.POINTER(C.c_int32)
_c_i64p = C.POINTER(C.c_int64)
_c_f64p = C.POINTER(C.c_double)

# name -> (restype, argtypes); mirrors include/gmmhmm.h one to one
SIGNATURES = {
    "gh_last_error": (C.c_char_p, []),
    "gh_version": (C.c_int, []),
    "gh_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "gh_ctx_destroy": (None, [C.c_void_p]),
    "gh_ctx_sync": (C.c_int, [C.c_void_p]),
    "gh_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "gh_device_count": (C.c_int, []),
    "gh_event_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "gh_event_destroy": (None, [C.c_void_p]),
    "gh_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gh_ctx_wait_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gh_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "gh_gmm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _c_f64p, _c_f64p, _c_f64p,
                                C.POINTER(C.c_void_p)]),
    "gh_gmm_destroy": (None, [C.c_void_p]),
    "gh_batch_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, _c_i64p,
                                  C.POINTER(C.c_void_p)]),
    "gh_batch_create_wire": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, _c_i64p,
                                       C.POINTER(C.c_void_p)]),
    "gh_host_unpin": (C.c_int, [C.c_void_p]),
    "gh_batch_gather": (C.c_int, [C.c_void_p, C.c_void_p, _c_i64p, C.c_int64, C.c_int64, _c_i64p, C.POINTER(C.c_void_p)]),
    "gh_batch_jitter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_double]),
    "gh_batch_gather_runs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _c_i64p, _c_i64p, _c_i64p, C.c_int64, C.c_int64, _c_i64p,
                                       C.POINTER(C.c_void_p)]),
    "gh_batch_tile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "gh_device_sync": (C.c_int, [C.c_void_p]),
    "gh_ctx_last_chunks": (C.c_int, [C.c_void_p]),
    "gh_ctx_set_compat": (C.c_int, [C.c_void_p, C.c_int]),
    "gh_batch_wrap": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, _c_i64p,
                                C.POINTER(C.c_void_p)]),
    "gh_batch_destroy": (None, [C.c_void_p]),
    "gh_batch_create_from_cepstra": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, _c_f64p, _c_i64p,
                                               C.POINTER(C.c_void_p)]),
    "gh_batch_fetch_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gh_mfcc_frames": (C.c_int64, [C.c_int64, C.c_int, C.c_double]),
    "gh_mfcc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64,
                          C.c_void_p, _c_i64p, _c_i64p, _c_f64p, _c_f64p]),
    "gh_batch_create_from_pcm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                           C.c_double, C.c_double, C.c_int64, C.c_void_p, _c_i64p, _c_i64p,
                                           C.POINTER(C.c_void_p)]),
    "gh_loglik": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gh_loglik_sets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _c_i64p, _c_i32p, _c_i32p]),
    "gh_loglik_subset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _c_i32p, _c_i32p]),
    "gh_loglik_dev_ptr": (C.c_void_p, [C.c_void_p]),
    "gh_loglik_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gh_component_loglik": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, _c_f64p, _c_f64p]),
    "gh_distance_matrix": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, _c_f64p, _c_f64p, _c_f64p,
                                     C.c_int, _c_f64p]),
    "gh_lattices_create": (C.c_int, [C.c_void_p, C.c_int, _c_i64p, _c_i32p, _c_i64p, _c_i32p, _c_i32p,
                                     _c_f64p, _c_i64p, _c_i32p, _c_i64p, _c_i32p, C.POINTER(C.c_void_p)]),
    "gh_lattices_create_transcripts": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _c_f64p, _c_i32p, C.c_int64, _c_i64p, _c_i32p,
                                                 C.POINTER(C.c_void_p)]),
    "gh_lattices_destroy": (None, [C.c_void_p]),
    "gh_lattices_set_beam": (C.c_int, [C.c_void_p, C.c_int]),
    "gh_lattices_forms": (C.c_int, [C.c_void_p]),
    "gh_viterbi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _c_i32p, _c_f64p, _c_i32p, _c_i32p, _c_i64p,
                             _c_i32p, _c_f64p, _c_i64p]),
    "gh_viterbi_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, _c_f64p, _c_i32p, _c_i32p, _c_i64p,
                                   _c_i32p, _c_f64p, _c_i64p]),
    "gh_ctx_last_fused": (C.c_int, [C.c_void_p]),
    "gh_viterbi_path_cap": (C.c_int64, [C.c_void_p, C.c_int, C.c_int64]),
    "gh_viterbi_labels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _c_i32p, _c_i32p, _c_f64p, _c_i32p, _c_i32p,
                                    _c_i64p, _c_i32p]),
    "gh_align_segments": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _c_i32p, _c_f64p, _c_i32p, _c_i32p]),
    "gh_align_runs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _c_i32p, _c_f64p, _c_i32p, C.c_int, _c_i32p, _c_i32p]),
    "gh_viterbi_labels_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _c_i32p, _c_i32p, C.c_int, _c_f64p, _c_i32p,
                                           _c_i32p, C.c_int64, _c_i32p]),
    "gh_dtw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _c_f64p, _c_f64p, _c_f64p, C.c_int, _c_f64p,
                         _c_f64p, _c_i32p, _c_i32p]),
    "gh_kmeans_assign": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, _c_f64p, _c_f64p,
                                   _c_i32p]),
    "gh_kmeans_resident_clusters": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _c_i32p]),
    "gh_kmeans_assign_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _c_i64p, C.POINTER(C.c_uint8), C.c_int, _c_f64p,
                                         _c_f64p, _c_i32p, _c_i32p, _c_f64p]),
    "gh_em_accumulate_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _c_i64p, C.POINTER(C.c_uint8), C.c_int, _c_f64p,
                                         _c_f64p, _c_f64p, _c_f64p, _c_f64p, C.c_void_p]),
    "gh_forward_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _c_i32p, C.c_int, _c_f64p, _c_f64p, _c_f64p,
                                      _c_f64p, _c_i64p, _c_f64p, _c_f64p]),
    "gh_bw_accumulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, _c_f64p, C.c_void_p]),
    "gh_em_accumulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, _c_f64p, _c_f64p,
                                   _c_f64p, _c_f64p, _c_f64p]),
    "gh_gmm_update": (C.c_int, [C.c_void_p, C.c_void_p, _c_f64p, _c_f64p, _c_f64p]),
    "gh_em_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, _c_f64p, _c_f64p, _c_f64p, _c_f64p, _c_i32p,
                               C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_void_p)]),
    "gh_em_create_transcripts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, _c_f64p, _c_f64p, _c_f64p, _c_f64p,
                                           C.c_int64, _c_i64p, _c_i32p, _c_i32p, C.c_double, C.c_double, C.c_double, C.c_int,
                                           C.POINTER(C.c_void_p)]),
    "gh_em_destroy": (None, [C.c_void_p]),
    "gh_em_iteration": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _c_f64p]),
    "gh_em_iterations_done": (C.c_int, [C.c_void_p]),
    "gh_em_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "gh_em_phase_ms": (C.c_int, [C.c_void_p, C.c_void_p, _c_f64p]),
    "gh_em_history": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _c_f64p]),
    "gh_em_get_model": (C.c_int, [C.c_void_p, C.c_void_p, _c_f64p, _c_f64p, _c_f64p, _c_f64p]),
    "gh_em_packed": (C.c_int, [C.c_void_p, C.c_void_p, _c_f64p, _c_i64p]),
    "gh_fit_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _c_i64p, C.c_int, C.POINTER(C.c_void_p)]),
    "gh_fit_destroy": (None, [C.c_void_p]),
    "gh_fit_segment_means": (C.c_int, [C.c_void_p, C.c_void_p, _c_f64p]),
    "gh_fit_kmeans": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, _c_f64p, C.POINTER(C.c_uint8), C.c_int, C.c_int,
                                _c_f64p, _c_f64p, _c_f64p, _c_i32p]),
    "gh_fit_clusters": (C.c_int, [C.c_void_p, C.c_void_p, _c_i32p]),
    "gh_fit_set_ids": (C.c_int, [C.c_void_p, C.c_void_p, _c_i32p]),
    "gh_fit_dtw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _c_f64p, _c_f64p, _c_i32p, C.POINTER(C.c_uint8)]),
    "gh_fit_group_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint8), _c_f64p, _c_f64p, _c_f64p]),
    "gh_fit_em": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, _c_f64p, _c_f64p, _c_f64p, _c_f64p, _c_f64p, _c_f64p,
                            _c_f64p, C.c_int, C.c_int, _c_i32p]),
    "gh_comm_unique_id": (C.c_int, [C.c_char_p]),
    "gh_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]),
    "gh_comm_destroy": (None, [C.c_void_p]),
    "gh_comm_abort": (C.c_int, [C.c_void_p]),
    "gh_comm_count": (C.c_int, [C.c_void_p]),
    "gh_comm_rank": (C.c_int, [C.c_void_p]),
    "gh_comm_version": (C.c_int, []),
    "gh_comm_library": (C.c_char_p, []),
    "gh_stats_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "gh_comm_allreduce_host": (C.c_int, [C.c_void_p, C.c_void_p, _c_f64p, C.c_int64, C.c_int]),
    "gh_comm_barrier": (C.c_int, [C.c_void_p, C.c_void_p]),
}


RESIDENT = object()   # kmeans_assign_multi(clusters=RESIDENT): assignments stay on the device between calls


class BackendError(RuntimeError):
    """The HIP backend is unavailable or a library call failed."""


class CommError(BackendError):
    """A collective failed or ran into its deadline (GH_ERR_COMM): a peer rank is gone; the communicator has been aborted."""


_lib = None
_lock = threading.Lock()


def load_library(path=None):
    """dlopen libgmmhmm.so and declare every prototype.  Needs no GPU."""
    global _lib
    with _lock:
        if _lib is not None and path is None:
            return _lib
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise BackendError(
                "libgmmhmm.so not found at %s -- build it with `python speech-recognition_amd/build.py` "
                "(there is no CPU fallback)" % p)
        try:
            lib = C.CDLL(p)
        except OSError as e:
            raise BackendError("cannot load %s: %s" % (p, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if path is None:
            _lib = lib
        return lib


def host_unpin(array):
    """End the page-locking a `Batch(..., pin="keep")` upload left on `array` (gh_host_unpin) -- before the array is freed."""
    load_library().gh_host_unpin(array.ctypes.data_as(C.c_void_p))


def device_count():
    """GPUs visible to the library's HIP runtime (0 when the library or a GPU is missing)."""
    try:
        return max(0, int(load_library().gh_device_count()))
    except BackendError:
        return 0


def _check(lib, rc):
    if rc != 0:
        msg = lib.gh_last_error().decode("utf-8", "replace")
        if rc == GH_ERR_SELF_POINTER:
            raise NameError("FUCKED")  # the reference's own exception (decode.py:120-121)
        if rc == GH_ERR_COMM:
            raise CommError("libgmmhmm error %d: %s" % (rc, msg))
        raise BackendError("libgmmhmm error %d: %s" % (rc, msg))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


class Context:
    """One GPU (gh_ctx)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib, self.lib.gh_ctx_create(int(device), C.byref(h)))
        self.h = h
        self.device = int(device)

    def sync(self):
        _check(self.lib, self.lib.gh_ctx_sync(self.h))

    def set_compat(self, underflow=True, lse_f32=False):
        """underflow=True (a context's default): likelihoods of states whose every weighted density underflows fp64 come
        back as +inf, like the reference's linear-domain GMM.evaluate (hmm_state.py:114-120); False: log domain throughout,
        finite costs.  lse_f32=True: the fp64 likelihood kernel takes the exponentials of its log-sum-exp in fp32
        (|delta nll| <= ~2.4e-7 absolute; faster).  See gh_ctx_set_compat."""
        _check(self.lib, self.lib.gh_ctx_set_compat(self.h, (1 if underflow else 0) | (2 if lse_f32 else 0)))

    @property
    def last_fused(self):
        """True when the last fused decode on this context ran the fused kernel (False: gh_loglik + gh_viterbi)."""
        return int(self.lib.gh_ctx_last_fused(self.h)) == 1

    @property
    def last_chunks(self):
        """Launches the last viterbi / forward_backward call on this context was cut into (scratch budget)."""
        return int(self.lib.gh_ctx_last_chunks(self.h))

    def device_sync(self):
        """hipDeviceSynchronize through the library's runtime: every context's stream on this GPU has drained."""
        _check(self.lib, self.lib.gh_device_sync(self.h))

    @property
    def stream(self):
        return self.lib.gh_ctx_stream(self.h)

    # hipEvent timing on this context's stream, through the library's own HIP runtime
    def new_event(self):
        e = C.c_void_p()
        _check(self.lib, self.lib.gh_event_create(self.h, C.byref(e)))
        return e

    def record(self, event):
        _check(self.lib, self.lib.gh_event_record(self.h, event))

    def wait_event(self, event):
        """Work submitted to this context after the call starts only when `event` (of any context) has completed."""
        _check(self.lib, self.lib.gh_ctx_wait_event(self.h, event))

    def elapsed_ms(self, start, stop):
        ms = C.c_float()
        _check(self.lib, self.lib.gh_event_elapsed_ms(start, stop, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gh_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=None):
    if device is None:
        device = int(os.environ.get("GMMHMM_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = _default_ctx[device] = Context(device)
    return ctx


COMM_ID_BYTES = 128


class Comm:
    """RCCL communicator of this rank on a context's GPU (gh_comm): the collectives run on the context's stream."""

    @staticmethod
    def unique_id():
        """The 128-byte id rank 0 makes and hands to every other rank (gh_comm_unique_id)."""
        lib = load_library()
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _check(lib, lib.gh_comm_unique_id(buf))
        return buf.raw

    def __init__(self, ctx, rank, world, unique_id):
        assert len(unique_id) == COMM_ID_BYTES
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        h = C.c_void_p()
        _check(ctx.lib, ctx.lib.gh_comm_create(ctx.h, self.rank, self.world, unique_id, C.byref(h)))
        self.h = h

    @property
    def count(self):
        """ncclCommCount: the number of ranks RCCL reports for this communicator."""
        return int(self.ctx.lib.gh_comm_count(self.h))

    def allreduce_device(self, dev_ptr, n):
        """In-place fp64 sum of n doubles at device pointer dev_ptr over all ranks, asynchronous on the context's stream."""
        _check(self.ctx.lib, self.ctx.lib.gh_stats_allreduce(self.ctx.h, self.h, C.c_void_p(int(dev_ptr)), int(n)))

    def allreduce_host(self, a, op="sum"):
        """Sum (or max) of a small fp64 host array over all ranks; returns a new array of the same shape."""
        out = np.array(a, dtype=np.float64, order="C")
        _check(self.ctx.lib, self.ctx.lib.gh_comm_allreduce_host(self.ctx.h, self.h, _ptr(out.reshape(-1), _c_f64p), out.size,
                                                                 1 if op == "max" else 0))
        return out

    def barrier(self):
        _check(self.ctx.lib, self.ctx.lib.gh_comm_barrier(self.ctx.h, self.h))

    def abort(self):
        """Failure path (gh_comm_abort): this rank leaves its collectives; what a rank does before it exits on an error,
        so that its peers' connections close instead of waiting for it."""
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.gh_comm_abort(self.h)

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                self.ctx.lib.gh_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PackedGMM:
    """S mixtures x M components x D dims on the GPU (gh_gmm)."""

    def __init__(self, ctx, mean, var, weight):
        mean, var, weight = _f64(mean), _f64(var), _f64(weight)
        S, M, D = mean.shape
        assert var.shape == (S, M, D) and weight.shape == (S, M)
        self.ctx, self.S, self.M, self.D = ctx, S, M, D
        h = C.c_void_p()
        rc = ctx.lib.gh_gmm_create(ctx.h, S, M, D, _ptr(mean, _c_f64p), _ptr(var, _c_f64p),
                                   _ptr(weight, _c_f64p), C.byref(h))
        if rc == -1 and b"singular" in ctx.lib.gh_last_error():
            raise np.linalg.LinAlgError("Singular matrix")  # hmm_state.py:17,30
        _check(ctx.lib, rc)
        self.h = h

    def update(self, mean, var, weight):
        """New parameters of the same shape, packed by kernels into the arrays the handle already owns (gh_gmm_update)."""
        mean, var, weight = _f64(mean), _f64(var), _f64(weight)
        assert mean.shape == (self.S, self.M, self.D) and var.shape == mean.shape and weight.shape == (self.S, self.M)
        rc = self.ctx.lib.gh_gmm_update(self.ctx.h, self.h, _ptr(mean, _c_f64p), _ptr(var, _c_f64p), _ptr(weight, _c_f64p))
        if rc == -1 and b"singular" in self.ctx.lib.gh_last_error():
            raise np.linalg.LinAlgError("Singular matrix")
        _check(self.ctx.lib, rc)

    def component_loglik(self, state, x):
        x = _f64(x)
        out = np.empty((x.shape[0], self.M))
        _check(self.ctx.lib, self.ctx.lib.gh_component_loglik(self.ctx.h, self.h, int(state), x.shape[0],
                                                              _ptr(x, _c_f64p), _ptr(out, _c_f64p)))
        return out

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):  # a handle must not outlive its context (interpreter shutdown order)
                self.ctx.lib.gh_gmm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Unsupported(BackendError):
    """The shapes are outside what a device-resident fast path covers; the caller keeps the general path."""


class EMSession:
    """Device-resident soft-EM (gh_em): `iteration()` enqueues likelihoods -> forward-backward -> statistics ->
    [all-reduce] -> M-step -> model re-pack on the context's stream.  utt_word: the word of every utterance (one-word
    transcripts, gh_em_create) -- or, with `transcripts` (a list of distinct label strings), the index of every
    utterance's transcript in that list (word strings, gh_em_create_transcripts)."""

    def __init__(self, ctx, batch, means, vars_, weights, word_trans, utt_word, var_floor, occ_floor=0.0,
                 min_occupancy=1e-8, update_transitions=True, transcripts=None):
        means, vars_, weights = _f64(means), _f64(vars_), _f64(weights)
        wt = _f64(word_trans)
        W, n = wt.shape[0], wt.shape[1]
        S, M, D = means.shape
        assert S == W * n and wt.shape == (W, n, n) and vars_.shape == means.shape and weights.shape == (S, M) and D == batch.D
        uw = np.ascontiguousarray(utt_word, dtype=np.int32)
        assert len(uw) == batch.U
        self.ctx, self.batch, self.W, self.n, self.S, self.M, self.D = ctx, batch, W, n, S, M, D
        h = C.c_void_p()
        self.word_strings = transcripts is not None
        if transcripts is not None:
            label_off, labels = Lattices.flatten_transcripts(transcripts)
            assert len(labels) == 0 or (labels.min() >= 0 and labels.max() < W), "transcripts: word index out of range"
            rc = ctx.lib.gh_em_create_transcripts(ctx.h, batch.h, W, n, M, _ptr(means, _c_f64p), _ptr(vars_, _c_f64p),
                                                  _ptr(weights, _c_f64p), _ptr(wt, _c_f64p), len(transcripts),
                                                  _ptr(label_off, _c_i64p), _ptr(labels, _c_i32p), _ptr(uw, _c_i32p),
                                                  float(var_floor), float(occ_floor), float(min_occupancy),
                                                  int(bool(update_transitions)), C.byref(h))
        else:
            rc = ctx.lib.gh_em_create(ctx.h, batch.h, W, n, M, _ptr(means, _c_f64p), _ptr(vars_, _c_f64p), _ptr(weights, _c_f64p),
                                      _ptr(wt, _c_f64p), _ptr(uw, _c_i32p), float(var_floor), float(occ_floor),
                                      float(min_occupancy), int(bool(update_transitions)), C.byref(h))
        if rc == GH_ERR_UNSUPPORTED:
            raise Unsupported(ctx.lib.gh_last_error().decode("utf-8", "replace"))
        if rc == -1 and b"singular" in ctx.lib.gh_last_error():
            raise np.linalg.LinAlgError("Singular matrix")
        _check(ctx.lib, rc)
        self.h = h
        batch.S = S

    def iteration(self, comm=None, sync=True):
        """One EM iteration; comm: a `Comm` whose ranks' statistics are summed before the M-step.  sync=True returns
        (log P before the update, utterances, converged); sync=False returns None as soon as the work is enqueued."""
        tail = np.empty(4) if sync else None
        rc = self.ctx.lib.gh_em_iteration(self.ctx.h, self.h, None if comm is None else comm.h, _ptr(tail, _c_f64p))
        if rc == -1 and b"singular" in self.ctx.lib.gh_last_error():
            raise np.linalg.LinAlgError("Singular matrix")
        _check(self.ctx.lib, rc)
        return None if tail is None else (float(tail[0]), float(tail[1]), bool(tail[2]))

    def profile(self, on=True):
        """HIP events between the phases of every following iteration (gh_em_profile)."""
        _check(self.ctx.lib, self.ctx.lib.gh_em_profile(self.ctx.h, self.h, 1 if on else 0))

    def phase_ms(self):
        """[likelihoods, forward-backward, statistics, tail + collective + M-step + re-pack] of the last profiled
        iteration, in milliseconds (gh_em_phase_ms; waits for it)."""
        out = np.zeros(4)
        _check(self.ctx.lib, self.ctx.lib.gh_em_phase_ms(self.ctx.h, self.h, _ptr(out, _c_f64p)))
        return out

    @property
    def iterations_done(self):
        return int(self.ctx.lib.gh_em_iterations_done(self.h))

    def history(self, first=0, count=None):
        """[count, 4] rows (log P before the update, utterances, converged, error bits) of iterations first .."""
        count = self.iterations_done - first if count is None else count
        out = np.empty((count, 4))
        _check(self.ctx.lib, self.ctx.lib.gh_em_history(self.ctx.h, self.h, int(first), int(count), _ptr(out, _c_f64p)))
        return out

    def model(self):
        """(means [S,M,D], vars [S,M,D], weights [S,M], word transition costs [W,n,n]) copied from the device."""
        mean, var = np.empty((self.S, self.M, self.D)), np.empty((self.S, self.M, self.D))
        w, t = np.empty((self.S, self.M)), np.empty((self.W, self.n, self.n))
        _check(self.ctx.lib, self.ctx.lib.gh_em_get_model(self.ctx.h, self.h, _ptr(mean, _c_f64p), _ptr(var, _c_f64p),
                                                          _ptr(w, _c_f64p), _ptr(t, _c_f64p)))
        return mean, var, w, t

    def packed(self):
        """The buffer that crosses the ranks, as the last iteration left it: [statistics | self transitions | log P | utterances]."""
        n = C.c_int64()
        _check(self.ctx.lib, self.ctx.lib.gh_em_packed(self.ctx.h, self.h, None, C.byref(n)))
        out = np.empty(n.value)
        _check(self.ctx.lib, self.ctx.lib.gh_em_packed(self.ctx.h, self.h, _ptr(out, _c_f64p), C.byref(n)))
        return out

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                self.ctx.lib.gh_em_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FitSession:
    """Device-resident refit of all states (gh_fit): frames of state s = rows [seg_off[s], seg_off[s+1]) of `batch`."""

    available = True

    def __init__(self, ctx, batch, seg_off, kmax):
        self.ctx, self.batch = ctx, batch
        self.seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
        self.S, self.D, self.N, self.kmax = len(self.seg_off) - 1, batch.D, batch.N, int(kmax)
        h = C.c_void_p()
        rc = ctx.lib.gh_fit_create(ctx.h, batch.h, self.S, _ptr(self.seg_off, _c_i64p), self.kmax, C.byref(h))
        if rc == GH_ERR_UNSUPPORTED:
            raise Unsupported(ctx.lib.gh_last_error().decode("utf-8", "replace"))
        _check(ctx.lib, rc)
        self.h = h

    def segment_means(self):
        """(sums [S, D] in frame order, counts [S]): sums / counts is np.mean(segment, axis=0), bit for bit."""
        out = np.empty((self.S, self.D + 1))
        _check(self.ctx.lib, self.ctx.lib.gh_fit_segment_means(self.ctx.h, self.h, _ptr(out, _c_f64p)))
        return out[:, :self.D], out[:, self.D]

    def kmeans(self, k, centroids, part, max_iteration=1000, check_every=8, comm=None):
        """-> (centroids [S,k,D], cov [S,k,D], counts [S,k], iterations [S])"""
        centroids = _f64(centroids)
        assert centroids.shape == (self.S, k, self.D)
        part = np.ascontiguousarray(part, dtype=np.uint8)
        assert part.shape == (self.N,)
        cen, cov = np.empty((self.S, k, self.D)), np.empty((self.S, k, self.D))
        cnt, its = np.empty((self.S, k)), np.empty(self.S, dtype=np.int32)
        _check(self.ctx.lib, self.ctx.lib.gh_fit_kmeans(
            self.ctx.h, self.h, None if comm is None else comm.h, int(k), _ptr(centroids, _c_f64p),
            part.ctypes.data_as(C.POINTER(C.c_uint8)), int(max_iteration), int(check_every), _ptr(cen, _c_f64p), _ptr(cov, _c_f64p),
            _ptr(cnt, _c_f64p), _ptr(its, _c_i32p)))
        return cen, cov, cnt, its

    def set_ids(self, ids):
        """The group (segment / cluster) of every frame, int32 [N]."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        assert ids.shape == (self.N,)
        _check(self.ctx.lib, self.ctx.lib.gh_fit_set_ids(self.ctx.h, self.h, _ptr(ids, _c_i32p)))

    def dtw(self, n, y, trans, utt_model, active=None):
        """dtw of every utterance against the n template rows y[utt_model[u]] under trans[utt_model[u]] (gh_fit_dtw); the
        row every frame is aligned to stays on the device as the session's ids."""
        y, trans = _f64(y), _f64(trans)
        assert y.shape == (self.S, n, self.D) and trans.shape == (self.S, n, n)
        um = np.ascontiguousarray(utt_model, dtype=np.int32)
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        _check(self.ctx.lib, self.ctx.lib.gh_fit_dtw(self.ctx.h, self.h, int(n), _ptr(y, _c_f64p), _ptr(trans, _c_f64p), _ptr(um, _c_i32p),
                                                     None if act is None else act.ctypes.data_as(C.POINTER(C.c_uint8))))

    def group_stats(self, k, active=None):
        """(mean [S,k,D], var [S,k,D] (ddof 1), count [S,k]) of the frames of every (state, group) from the resident ids."""
        mean, var, cnt = np.empty((self.S, k, self.D)), np.empty((self.S, k, self.D)), np.empty((self.S, k))
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        _check(self.ctx.lib, self.ctx.lib.gh_fit_group_stats(self.ctx.h, self.h, int(k),
                                                             None if act is None else act.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                             _ptr(mean, _c_f64p), _ptr(var, _c_f64p), _ptr(cnt, _c_f64p)))
        return mean, var, cnt

    def clusters(self):
        out = np.empty(self.N, dtype=np.int32)
        _check(self.ctx.lib, self.ctx.lib.gh_fit_clusters(self.ctx.h, self.h, _ptr(out, _c_i32p)))
        return out

    def em(self, k, mean, var, weight, mu_old, sigma_old, w_old, n_frames, max_iteration=10000, check_every=8, comm=None):
        """In-place on the six [S,k,..] arrays (float64, C-contiguous); returns converged_at [S] (-1: not converged)."""
        for a, shp in ((mean, (self.S, k, self.D)), (var, (self.S, k, self.D)), (weight, (self.S, k)), (mu_old, (self.S, k, self.D)),
                       (sigma_old, (self.S, k, self.D)), (w_old, (self.S, k))):
            assert a.dtype == np.float64 and a.flags.c_contiguous and a.shape == shp
        nf = _f64(n_frames)
        conv = np.empty(self.S, dtype=np.int32)
        rc = self.ctx.lib.gh_fit_em(self.ctx.h, self.h, None if comm is None else comm.h, int(k), _ptr(mean, _c_f64p),
                                    _ptr(var, _c_f64p), _ptr(weight, _c_f64p), _ptr(mu_old, _c_f64p), _ptr(sigma_old, _c_f64p),
                                    _ptr(w_old, _c_f64p), _ptr(nf, _c_f64p), int(max_iteration), int(check_every), _ptr(conv, _c_i32p))
        if rc == -1 and b"singular" in self.ctx.lib.gh_last_error():
            raise np.linalg.LinAlgError("Singular matrix")
        _check(self.ctx.lib, rc)
        return conv

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                self.ctx.lib.gh_fit_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_PCM_FMT = {np.dtype(np.int16): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}


def pack_pcm(ctx, signals, sample_rate, mfcc_params=None):
    """Concatenate audio signals for gh_mfcc / gh_batch_create_from_pcm.  int16 / float32 / float64 travel as they
    are, anything else as float64.  mfcc_params = (frame_size, frame_stride, low_freq, high_freq), defaults of
    mfcc_features (feature.py:43): 0.025, 0.01, 80, None.  Returns (samples, fmt, sample_off, frame_off, params)."""
    fs, st, lo, hi = mfcc_params if mfcc_params is not None else (0.025, 0.01, 80, None)
    prm = (float(fs), float(st), float(lo), 0.0 if hi is None else float(hi))
    sigs = [np.asarray(x).reshape(-1) for x in signals]
    dt = sigs[0].dtype if sigs and all(x.dtype == sigs[0].dtype for x in sigs) and sigs[0].dtype in _PCM_FMT \
        else np.dtype(np.float64)
    s_off = np.zeros(len(sigs) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in sigs], out=s_off[1:])
    for x in sigs:
        if len(x) == 0:
            raise IndexError("index 0 is out of bounds for axis 0 with size 0")  # signal[0], feature.py:46
    samples = np.ascontiguousarray(np.concatenate(sigs).astype(dt, copy=False)) if sigs else np.zeros(0, dtype=dt)
    f_off = np.zeros(len(sigs) + 1, dtype=np.int64)
    np.cumsum([ctx.lib.gh_mfcc_frames(len(x), int(sample_rate), prm[1]) for x in sigs], out=f_off[1:])
    return samples, _PCM_FMT[dt], s_off, f_off, prm


def mfcc(ctx, signals, sample_rate=16000, mfcc_params=None):
    """mfcc_features (feature.py:43-82) for a list of signals: ([T_u,40] log10 filterbank energies, [T_u,13] cepstra)."""
    samples, fmt, s_off, f_off, prm = pack_pcm(ctx, signals, sample_rate, mfcc_params)
    N = int(f_off[-1])
    fb, mf = np.empty((N, 40)), np.empty((N, 13))
    _check(ctx.lib, ctx.lib.gh_mfcc(ctx.h, fmt, int(sample_rate), prm[0], prm[1], prm[2], prm[3], len(s_off) - 1,
                                    samples.ctypes.data_as(C.c_void_p), _ptr(s_off, _c_i64p), _ptr(f_off, _c_i64p),
                                    _ptr(fb, _c_f64p), _ptr(mf, _c_f64p)))
    U = len(s_off) - 1
    return ([fb[f_off[u]:f_off[u + 1]] for u in range(U)], [mf[f_off[u]:f_off[u + 1]] for u in range(U)])


def concat_rows(arrays, out):
    """np.concatenate(arrays, out=out) for the caller's templates: C-contiguous float64 arrays go through the threaded copy of
    `_hostcopy` (csrc/hostcopy.c: GMMHMM_HOST_THREADS threads, the GIL released; 62 MB in 2 000 pieces: 5.7 -> ~1.5 ms), anything
    else -- other dtypes, strided views, lists -- through numpy with its casts."""
    import os
    try:
        from . import _hostcopy
    except ImportError:
        _hostcopy = None
    if _hostcopy is not None:
        try:
            _hostcopy.concat_rows(arrays, out, min(int(os.environ.get("GMMHMM_HOST_THREADS", "8")), os.cpu_count() or 1))
            return out
        except TypeError:
            pass
    D = out.shape[1]
    try:
        np.concatenate(arrays, out=out)                            # [T, D] arrays: no per-template call
    except (ValueError, TypeError):
        np.concatenate([np.asarray(t).reshape(-1, D) for t in arrays], out=out)
    return out


class Batch:
    """Ragged batch of utterances resident in HBM (gh_batch)."""

    def __init__(self, ctx, utterances=None, dtype=np.float64, feats=None, offsets=None, cepstra=None, frontend_mode=0,
                 pcm=None, sample_rate=16000, mfcc_params=None, feats_dev=None, dim=None, wire=None, pin=False):
        """wire=np.float32 with dtype float64: `feats` cross the host link as fp32 and are widened on the device
        (gh_batch_create_wire); pin=True page-locks the host buffer for the copy, pin="keep" leaves it page-locked for
        the next upload from the same array (`host_unpin(array)` ends that, before the array is freed)."""
        self.ctx = ctx
        self.np_dtype = np.dtype(dtype)
        assert self.np_dtype in (np.dtype(np.float32), np.dtype(np.float64))
        if feats_dev is not None:  # features already in HBM (gh_batch_wrap): device pointer of a row-major [N, dim] matrix
            self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
            self.N, self.D, self.U = int(self.offsets[-1]), int(dim), len(self.offsets) - 1
            h = C.c_void_p()
            _check(ctx.lib, ctx.lib.gh_batch_wrap(ctx.h, GH_F64 if self.np_dtype == np.float64 else GH_F32, self.D,
                                                  self.N, self.U, C.c_void_p(int(feats_dev)),
                                                  _ptr(self.offsets, _c_i64p), C.byref(h)))
            self.h = h
            self.S = None
            return
        if pcm is not None:  # N3 front-end from audio samples: MFCC -> [ceps | delta | delta-delta] -> standardise
            samples, fmt, s_off, f_off, prm = pack_pcm(ctx, pcm, sample_rate, mfcc_params)
            for n in np.diff(f_off):
                if n < 2 and frontend_mode != 2:
                    raise IndexError("index 1 is out of bounds for axis 0 with size %d" % n)  # core.py:16
            self.offsets = f_off
            self.N, self.D, self.U = int(f_off[-1]), (13 if frontend_mode == 2 else 39), len(f_off) - 1
            h = C.c_void_p()
            _check(ctx.lib, ctx.lib.gh_batch_create_from_pcm(
                ctx.h, GH_F64 if self.np_dtype == np.float64 else GH_F32, int(frontend_mode), fmt, int(sample_rate),
                prm[0], prm[1], prm[2], prm[3], self.U, samples.ctypes.data_as(C.c_void_p), _ptr(s_off, _c_i64p),
                _ptr(f_off, _c_i64p), C.byref(h)))
            self.h = h
            self.S = None
            return
        if cepstra is not None:  # N3 front-end: [ceps | delta | delta-delta], standardised per utterance, on the GPU
            lens = [len(u) for u in cepstra]
            for n in lens:
                if n < 2 and frontend_mode != 2:
                    raise IndexError("index 1 is out of bounds for axis 0 with size %d" % n)  # core.py:16
            self.offsets = np.zeros(len(lens) + 1, dtype=np.int64)
            np.cumsum(lens, out=self.offsets[1:])
            Cc = np.asarray(cepstra[0]).shape[1] if lens else 1
            ceps = np.ascontiguousarray(np.concatenate([np.asarray(u, dtype=np.float64).reshape(-1, Cc) for u in cepstra])
                                        if lens else np.zeros((0, Cc)))
            self.N, self.D, self.U = ceps.shape[0], (Cc if frontend_mode == 2 else 3 * Cc), len(lens)
            h = C.c_void_p()
            _check(ctx.lib, ctx.lib.gh_batch_create_from_cepstra(
                ctx.h, GH_F64 if self.np_dtype == np.float64 else GH_F32, int(frontend_mode), Cc, self.N, self.U,
                _ptr(ceps, _c_f64p),
                _ptr(self.offsets, _c_i64p), C.byref(h)))
            self.h = h
            self.S = None
            return
        if utterances is not None:
            lens = [len(u) for u in utterances]
            offsets = np.zeros(len(lens) + 1, dtype=np.int64)
            np.cumsum(lens, out=offsets[1:])
            D = np.asarray(utterances[0]).shape[1] if lens else 1
            if lens and self.np_dtype == np.float64:
                feats = concat_rows(utterances, np.empty((int(offsets[-1]), D)))      # (threaded for float64 arrays)
            else:
                feats = (np.concatenate([np.asarray(u, dtype=self.np_dtype).reshape(-1, D) for u in utterances])
                         if lens else np.zeros((0, D), dtype=self.np_dtype))
        wire_dt = self.np_dtype if wire is None else np.dtype(wire)
        feats = np.ascontiguousarray(feats, dtype=wire_dt)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.N, self.D = feats.shape
        self.U = len(self.offsets) - 1
        h = C.c_void_p()
        code = lambda dt: GH_F64 if dt == np.float64 else GH_F32
        if wire is None and not pin:
            _check(ctx.lib, ctx.lib.gh_batch_create(ctx.h, code(self.np_dtype), self.D,
                                                    self.N, self.U, feats.ctypes.data_as(C.c_void_p),
                                                    _ptr(self.offsets, _c_i64p), C.byref(h)))
        else:
            _check(ctx.lib, ctx.lib.gh_batch_create_wire(ctx.h, code(self.np_dtype), code(wire_dt), 2 if pin == "keep" else (1 if pin else 0), self.D, self.N,
                                                         self.U, feats.ctypes.data_as(C.c_void_p), _ptr(self.offsets, _c_i64p),
                                                         C.byref(h)))
        self.h = h
        self.S = None

    @property
    def lengths(self):
        return np.diff(self.offsets)

    def features(self):
        """The resident feature matrix, copied back: list of [T_u, D] arrays."""
        out = np.empty((self.N, self.D), dtype=self.np_dtype)
        _check(self.ctx.lib, self.ctx.lib.gh_batch_fetch_features(self.ctx.h, self.h, out.ctypes.data_as(C.c_void_p)))
        return [out[self.offsets[u]:self.offsets[u + 1]] for u in range(self.U)]

    def loglik(self, gmm, fetch=True, state_ranges=None, state_sets=None):
        """A3 for every frame x state; the [N,S] matrix stays resident for the DPs.
        state_ranges = (lo [U], hi [U]): only the states [lo[u], hi[u]) of utterance u are needed (forced alignment /
        EM with known transcripts); the other entries of the matrix are then unspecified.
        state_sets = (range_off [U+1], lo, hi): several ranges per utterance (the words of its transcript)."""
        if state_sets is not None:
            off = np.ascontiguousarray(state_sets[0], dtype=np.int64)
            lo = np.ascontiguousarray(state_sets[1], dtype=np.int32)
            hi = np.ascontiguousarray(state_sets[2], dtype=np.int32)
            assert len(off) == self.U + 1 and len(lo) == len(hi) == int(off[-1])
            _check(self.ctx.lib, self.ctx.lib.gh_loglik_sets(self.ctx.h, gmm.h, self.h, _ptr(off, _c_i64p), _ptr(lo, _c_i32p),
                                                             _ptr(hi, _c_i32p)))
            self.S = gmm.S
            if not fetch:
                return None
            out = np.empty((self.N, gmm.S), dtype=self.np_dtype)
            _check(self.ctx.lib, self.ctx.lib.gh_loglik_fetch(self.ctx.h, self.h, out.ctypes.data_as(C.c_void_p)))
            return out
        if state_ranges is not None:
            lo = np.ascontiguousarray(state_ranges[0], dtype=np.int32)
            hi = np.ascontiguousarray(state_ranges[1], dtype=np.int32)
            assert len(lo) == self.U and len(hi) == self.U
            _check(self.ctx.lib, self.ctx.lib.gh_loglik_subset(self.ctx.h, gmm.h, self.h, _ptr(lo, _c_i32p), _ptr(hi, _c_i32p)))
            self.S = gmm.S
            if not fetch:
                return None
            out = np.empty((self.N, gmm.S), dtype=self.np_dtype)
            _check(self.ctx.lib, self.ctx.lib.gh_loglik_fetch(self.ctx.h, self.h, out.ctypes.data_as(C.c_void_p)))
            return out
        out = np.empty((self.N, gmm.S), dtype=self.np_dtype) if fetch else None
        _check(self.ctx.lib, self.ctx.lib.gh_loglik(self.ctx.h, gmm.h, self.h,
                                                    None if out is None else out.ctypes.data_as(C.c_void_p)))
        self.S = gmm.S
        return out

    def dtw(self, trans, y=None, var=None, beam=0, dist=None, want_costs=True):
        """A5 for every utterance against the n template rows `y` (built-in Euclidean /
        mahalanobis distance) or caller-supplied distance matrices `dist` (list of [n,T_u]).
        Returns (costs list of [n,T_u] or None, paths list of int64 [K,2])."""
        lib = self.ctx.lib
        trans = _f64(trans)
        n = trans.shape[0]
        y = None if y is None else _f64(y)
        var = None if var is None else _f64(var)
        dflat = None
        if dist is not None:
            dflat = np.ascontiguousarray(np.concatenate([_f64(d).ravel() for d in dist]))
        costs = np.empty(n * self.N) if want_costs else None
        path = np.empty((self.N, 2), dtype=np.int32)
        plen = np.empty(self.U, dtype=np.int32)
        _check(lib, lib.gh_dtw(self.ctx.h, self.h, n, _ptr(y, _c_f64p), _ptr(var, _c_f64p), _ptr(trans, _c_f64p),
                               int(beam), _ptr(dflat, _c_f64p), _ptr(costs, _c_f64p), _ptr(path, _c_i32p),
                               _ptr(plen, _c_i32p)))
        off = self.offsets
        cl = None
        if want_costs:
            cl = [costs[n * off[u]:n * off[u + 1]].reshape(n, -1) for u in range(self.U)]
        return cl, [path[off[u]:off[u] + plen[u]].astype(np.int64) for u in range(self.U)]

    def kmeans_assign(self, centroids, var=None, first=0, count=None):
        """A14 assignment step over frames [first, first+count): index of the closest centroid."""
        centroids = _f64(centroids)
        count = self.N - first if count is None else count
        out = np.empty(count, dtype=np.int32)
        v = None if var is None else _f64(var)
        _check(self.ctx.lib, self.ctx.lib.gh_kmeans_assign(self.ctx.h, self.h, int(first), int(count),
                                                           centroids.shape[0], _ptr(centroids, _c_f64p),
                                                           _ptr(v, _c_f64p), _ptr(out, _c_i32p)))
        return out.astype(np.int64)

    def kmeans_assign_multi(self, seg_off, centroids, var=None, clusters=None, active=None, want_sums=False):
        """A14 for every state at once: frames [seg_off[s], seg_off[s+1]) belong to state s; centroids [S,k,D]; var [S,D]
        (mahalanobis under the state's shared variance) or None (Euclidean).  `clusters` (int32 [N]) is updated in
        place for the frames of active states.  Returns (clusters, changed [S], sums [S,k,D+1] or None)."""
        centroids = _f64(centroids)
        S, k, D = centroids.shape
        seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
        assert len(seg_off) == S + 1 and D == self.D
        if clusters is None:
            clusters = np.full(self.N, -1, dtype=np.int32)
        elif clusters is RESIDENT:
            clusters = None         # the assignments stay in the batch, on the device (see resident_clusters)
        assert clusters is None or (clusters.dtype == np.int32 and clusters.flags.c_contiguous and len(clusters) == self.N)
        v = None if var is None else _f64(var).reshape(S, D)
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        changed = np.zeros(S, dtype=np.int32)
        sums = np.zeros((S, k, D + 1)) if want_sums else None
        _check(self.ctx.lib, self.ctx.lib.gh_kmeans_assign_multi(
            self.ctx.h, self.h, S, _ptr(seg_off, _c_i64p), None if act is None else act.ctypes.data_as(C.POINTER(C.c_uint8)),
            k, _ptr(centroids, _c_f64p), _ptr(v, _c_f64p), _ptr(clusters, _c_i32p), _ptr(changed, _c_i32p),
            _ptr(sums, _c_f64p)))
        return clusters, changed, sums

    def gather(self, rows, offsets=None):
        """Rows `rows` of this batch as a new resident batch (gh_batch_gather: copied on the device); offsets: utterance
        boundaries of the new batch (default: one utterance)."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        off = np.ascontiguousarray([0, len(rows)] if offsets is None else offsets, dtype=np.int64)
        new = Batch.__new__(Batch)
        new.ctx, new.np_dtype = self.ctx, self.np_dtype
        new.offsets, new.N, new.D, new.U, new.S = off, len(rows), self.D, len(off) - 1, None
        h = C.c_void_p()
        _check(self.ctx.lib, self.ctx.lib.gh_batch_gather(self.ctx.h, self.h, _ptr(rows, _c_i64p), len(rows), new.U,
                                                          _ptr(off, _c_i64p), C.byref(h)))
        new.h = h
        return new

    def gather_runs(self, start, length, dest, n, offsets=None):
        """Runs of consecutive rows -- rows [start[r], start[r] + length[r]) of this batch become rows [dest[r], ...) of a new
        resident batch of n rows (gh_batch_gather_runs: one contiguous copy per run, 24 bytes of table per run instead of 8
        bytes of index per row)."""
        start, length, dest = (np.ascontiguousarray(a, dtype=np.int64) for a in (start, length, dest))
        off = np.ascontiguousarray([0, int(n)] if offsets is None else offsets, dtype=np.int64)
        new = Batch.__new__(Batch)
        new.ctx, new.np_dtype = self.ctx, self.np_dtype
        new.offsets, new.N, new.D, new.U, new.S = off, int(n), self.D, len(off) - 1, None
        h = C.c_void_p()
        _check(self.ctx.lib, self.ctx.lib.gh_batch_gather_runs(self.ctx.h, self.h, len(start), _ptr(start, _c_i64p),
                                                               _ptr(length, _c_i64p), _ptr(dest, _c_i64p), int(n), new.U,
                                                               _ptr(off, _c_i64p), C.byref(h)))
        new.h = h
        return new

    def jitter(self, seed, scale):
        """feats += scale * N(0, 1), independent per feature, generated on the device (gh_batch_jitter): the copies of a
        tiled batch become utterances of their own."""
        _check(self.ctx.lib, self.ctx.lib.gh_batch_jitter(self.ctx.h, self.h, int(seed) & (2 ** 64 - 1), float(scale)))
        return self

    def tile(self, reps):
        """`reps` copies of this batch back to back as a new resident batch (gh_batch_tile: device-to-device)."""
        new = Batch.__new__(Batch)
        new.ctx, new.np_dtype, new.D, new.S = self.ctx, self.np_dtype, self.D, None
        new.N, new.U = self.N * reps, self.U * reps
        T = np.tile(np.diff(self.offsets), reps)
        new.offsets = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
        h = C.c_void_p()
        _check(self.ctx.lib, self.ctx.lib.gh_batch_tile(self.ctx.h, self.h, int(reps), C.byref(h)))
        new.h = h
        return new

    def resident_clusters(self, reset=False, fetch=True):
        """The k-means assignments kept on the device by kmeans_assign_multi(clusters=RESIDENT): reset to -1 and / or
        fetch them (int32 [N])."""
        out = np.empty(self.N, dtype=np.int32) if fetch else None
        _check(self.ctx.lib, self.ctx.lib.gh_kmeans_resident_clusters(self.ctx.h, self.h, int(bool(reset)), _ptr(out, _c_i32p)))
        return out

    def em_accumulate_multi(self, seg_off, mean, var, weight, active=None, stats_dev=None):
        """A7 E-step statistics of every (active) state in one launch: mean / var [S,k,D], weight [S,k].
        Returns (stats [S,k,1+2D], loglik [S])."""
        mean, var, weight = _f64(mean), _f64(var), _f64(weight)
        S, k, D = mean.shape
        seg_off = np.ascontiguousarray(seg_off, dtype=np.int64)
        assert len(seg_off) == S + 1 and D == self.D and var.shape == (S, k, D) and weight.shape == (S, k)
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        stats = np.zeros((S, k, 1 + 2 * D))
        ll = np.zeros(S)
        rc = self.ctx.lib.gh_em_accumulate_multi(
            self.ctx.h, self.h, S, _ptr(seg_off, _c_i64p), None if act is None else act.ctypes.data_as(C.POINTER(C.c_uint8)),
            k, _ptr(mean, _c_f64p), _ptr(var, _c_f64p), _ptr(weight, _c_f64p), _ptr(stats, _c_f64p), _ptr(ll, _c_f64p),
            stats_dev)
        if rc == -1 and b"singular" in self.ctx.lib.gh_last_error():
            raise np.linalg.LinAlgError("Singular matrix")
        _check(self.ctx.lib, rc)
        return stats, ll

    def bw_accumulate(self, gmm, occ_floor=0.0, stats_dev=None, fetch=True):
        """Baum-Welch statistics [S, M, 1+2D] of the whole batch from the resident occupancies
        (run Lattices.forward_backward(..., want_occ=True) first).  stats_dev: device pointer (e.g. a torch
        tensor's data_ptr() that RCCL is about to all-reduce) that receives them; with fetch=False nothing is
        copied to the host and None is returned."""
        assert fetch or stats_dev, "bw_accumulate: nowhere to put the statistics"
        out = np.empty((gmm.S, gmm.M, 1 + 2 * self.D)) if fetch else None
        _check(self.ctx.lib, self.ctx.lib.gh_bw_accumulate(self.ctx.h, gmm.h, self.h, float(occ_floor),
                                                           _ptr(out, _c_f64p), stats_dev))
        return out

    def em_accumulate(self, mean, var, weight, first=0, count=None):
        """A7 E-step statistics of the given k components over frames [first, first+count).
        Returns (stats [k, 1+2D], total log-likelihood)."""
        mean, var, weight = _f64(mean), _f64(var), _f64(weight)
        k = mean.shape[0]
        count = self.N - first if count is None else count
        stats = np.empty((k, 1 + 2 * self.D))
        ll = C.c_double(0.0)
        rc = self.ctx.lib.gh_em_accumulate(self.ctx.h, self.h, int(first), int(count), k, _ptr(mean, _c_f64p),
                                           _ptr(var, _c_f64p), _ptr(weight, _c_f64p), _ptr(stats, _c_f64p),
                                           C.cast(C.byref(ll), _c_f64p))
        if rc == -1 and b"singular" in self.ctx.lib.gh_last_error():
            raise np.linalg.LinAlgError("Singular matrix")
        _check(self.ctx.lib, rc)
        return stats, ll.value

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):  # a handle must not outlive its context (interpreter shutdown order)
                self.ctx.lib.gh_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Lattices:
    """One or more DP graphs on the GPU (gh_lattices).

    `graphs` is a list of dicts with keys
        row_state [R] int (-1 = non-emitting), arc_to, arc_from [A] int, arc_cost [A] float,
        start_rows, end_rows (lists of local row indices)
    """

    def __init__(self, ctx, graphs):
        self.ctx = ctx
        L = len(graphs)
        cat = lambda key, dt: np.ascontiguousarray(
            np.concatenate([np.asarray(g[key], dtype=dt).ravel() for g in graphs]) if L else np.zeros(0, dt), dtype=dt)
        off = lambda key: np.ascontiguousarray(
            np.concatenate([[0], np.cumsum([len(g[key]) for g in graphs])]), dtype=np.int64)
        self.row_off, self.arc_off = off("row_state"), off("arc_to")
        self.start_off, self.end_off = off("start_rows"), off("end_rows")
        self.n_end = [len(g["end_rows"]) for g in graphs]
        self.R = [len(g["row_state"]) for g in graphs]
        row_state = cat("row_state", np.int32)
        arc_to, arc_from = cat("arc_to", np.int32), cat("arc_from", np.int32)
        arc_cost = cat("arc_cost", np.float64)
        start_rows, end_rows = cat("start_rows", np.int32), cat("end_rows", np.int32)
        self.end_rows = [np.asarray(g["end_rows"], dtype=np.int64) for g in graphs]
        h = C.c_void_p()
        _check(ctx.lib, ctx.lib.gh_lattices_create(
            ctx.h, L, _ptr(self.row_off, _c_i64p), _ptr(row_state, _c_i32p), _ptr(self.arc_off, _c_i64p),
            _ptr(arc_to, _c_i32p), _ptr(arc_from, _c_i32p), _ptr(arc_cost, _c_f64p),
            _ptr(self.start_off, _c_i64p), _ptr(start_rows, _c_i32p), _ptr(self.end_off, _c_i64p),
            _ptr(end_rows, _c_i32p), C.byref(h)))
        self.h = h
        self.L = L

    @staticmethod
    def flatten_transcripts(label_seqs):
        """(label_off int64 [L+1], labels int32 [sum K]) of a list of label strings -- what `from_transcripts` hands to
        the library; a trainer that rebuilds its graphs every iteration flattens once and passes `flat=`."""
        K = np.array([len(l) for l in label_seqs], dtype=np.int64)
        label_off = np.concatenate([[0], np.cumsum(K)]).astype(np.int64)
        import itertools
        labels = np.fromiter(itertools.chain.from_iterable(label_seqs), dtype=np.int32, count=int(label_off[-1]))
        return label_off, labels

    @classmethod
    def from_transcripts(cls, ctx, word_transitions, n, label_seqs=None, state_base=None, flat=None):
        """The forced-alignment graph of every label string in `label_seqs` (continuous_speech.py:80-82: one word per
        layer; the graphs `continuous_speech.packed_lattice(word_transitions, n, [[l] for l in labels])` describes),
        built on the library's side from the W [n, n] cost matrices -- see gh_lattices_create_transcripts."""
        self = cls.__new__(cls)
        self.ctx = ctx
        wt = np.ascontiguousarray(np.asarray([np.asarray(t, dtype=np.float64) for t in word_transitions]), dtype=np.float64)
        assert wt.ndim == 3 and wt.shape[1] == n and wt.shape[2] == n, "word_transitions: W matrices of n x n costs"
        label_off, labels = flat if flat is not None else cls.flatten_transcripts(label_seqs)
        K = np.diff(label_off)
        self.L = len(K)
        base = None if state_base is None else np.ascontiguousarray(state_base, dtype=np.int32)
        self.R = K * (n + 1) + 1
        self.n_end = np.ones(self.L, dtype=np.int64)
        self._n_per_word = n
        self._K = K
        h = C.c_void_p()
        _check(ctx.lib, ctx.lib.gh_lattices_create_transcripts(ctx.h, wt.shape[0], int(n), _ptr(wt, _c_f64p), _ptr(base, _c_i32p),
                                                               self.L, _ptr(label_off, _c_i64p), _ptr(labels, _c_i32p), C.byref(h)))
        self.h = h
        return self

    @property
    def end_rows(self):
        if "_end_rows" not in self.__dict__:      # (transcripts handle: the end row is the last state of the last word)
            n = self._n_per_word
            self._end_rows = [np.array([(k - 1) * (n + 1) + n], dtype=np.int64) for k in self._K]
        return self._end_rows

    @end_rows.setter
    def end_rows(self, value):
        self._end_rows = value

    def set_beam(self, beam):
        """Rank beam per column for viterbi / viterbi_labels (None, 0 or inf: no pruning); see gh_lattices_set_beam."""
        k = 0 if beam is None or beam != beam or beam == float("inf") or beam <= 0 else int(beam)
        _check(self.ctx.lib, self.ctx.lib.gh_lattices_set_beam(self.h, k))
        self.beam = k

    FORMS = ("chain", "layers", "loop", "sequence", "fb_chain")

    def forms(self):
        """Names of the special forms the graphs were recognised in (gh_lattices_forms): they select the kernels."""
        bits = int(self.ctx.lib.gh_lattices_forms(self.h))
        return {name for k, name in enumerate(self.FORMS) if bits >> k & 1}

    def path_cap(self, l, T):
        return int(self.ctx.lib.gh_viterbi_path_cap(self.h, int(l), int(T)))

    def viterbi(self, batch, utt_lattice=None, want_path=True, want_costs=False, want_end_cost=True, fused_gmm=None,
                log_domain=False, flat_paths=False):
        """A6 for every utterance.  Returns dict(end_cost [list per utt], best_end [U],
        paths [list of int64 [K,2]], costs [list of [R,T]]).  want_end_cost=False: only best_end (the cheapest end
        row, last minimum on ties: decode.py:129-134) comes back, the end costs stay on the device.
        fused_gmm: a one-component `PackedGMM` -- the cells are scored inside the sweep (gh_viterbi_fused; the batch
        needs no likelihood matrix); log_domain=True scores with mahalanobis() (no linear-domain underflow)."""
        lib, U = self.ctx.lib, batch.U
        lat = None if utt_lattice is None else np.ascontiguousarray(utt_lattice, dtype=np.int32)
        if lat is None and not (want_path or want_costs):
            # one graph, end costs / best ends only (the batch recognisers): no per-utterance bookkeeping on the host
            # (at 100 000 utterances the index arrays below cost as much wall time as a tenth of the sweep)
            n_end0 = int(self.n_end[0])
            end_cost = np.empty(U * n_end0, dtype=np.float64) if want_end_cost else None
            best_end = np.empty(U, dtype=np.int32)
            if fused_gmm is not None:
                _check(lib, lib.gh_viterbi_fused(self.ctx.h, fused_gmm.h, self.h, batch.h, int(bool(log_domain)),
                                                 _ptr(end_cost, _c_f64p), _ptr(best_end, _c_i32p), None, None, None, None, None))
            else:
                _check(lib, lib.gh_viterbi(self.ctx.h, self.h, batch.h, None, _ptr(end_cost, _c_f64p), _ptr(best_end, _c_i32p),
                                           None, None, None, None, None))
            end_off = np.arange(U + 1, dtype=np.int64) * n_end0 if (want_end_cost or U <= 2048) else None
            return dict(best_end=best_end, end_off=end_off, end_cost_flat=end_cost,
                        end_cost=[end_cost[end_off[u]:end_off[u + 1]] for u in range(U)] if (U <= 2048 and want_end_cost) else None)
        lidx = np.zeros(U, dtype=np.int64) if lat is None else lat.astype(np.int64)
        T = batch.lengths
        n_end = np.asarray(self.n_end, dtype=np.int64)[lidx]
        end_off = np.concatenate([[0], np.cumsum(n_end)])
        end_cost = np.empty(int(end_off[-1]), dtype=np.float64) if want_end_cost else None
        best_end = np.empty(U, dtype=np.int32)
        path = path_off = path_len = costs = costs_off = None
        if want_path:
            nlev = np.array([self.path_cap(l, 1) for l in range(self.L)], dtype=np.int64)     # (cells per column at most)
            cap = np.where(np.asarray(T) > 1, np.asarray(T, dtype=np.int64) * nlev[lidx], 0).astype(np.int64)
            path_off = np.concatenate([[0], np.cumsum(cap)]).astype(np.int64)
            path = np.empty((int(path_off[-1]), 2), dtype=np.int32)
            path_len = np.empty(U, dtype=np.int32)
        if want_costs:
            Rs = np.asarray(self.R, dtype=np.int64)[lidx]
            costs_off = np.concatenate([[0], np.cumsum(Rs * T)]).astype(np.int64)
            costs = np.empty(int(costs_off[-1]), dtype=np.float64)
        if fused_gmm is not None:
            assert lat is None, "the fused decode takes one graph for the whole batch"
            _check(lib, lib.gh_viterbi_fused(self.ctx.h, fused_gmm.h, self.h, batch.h, int(bool(log_domain)),
                                             _ptr(end_cost, _c_f64p), _ptr(best_end, _c_i32p), _ptr(path, _c_i32p),
                                             _ptr(path_off, _c_i64p), _ptr(path_len, _c_i32p), _ptr(costs, _c_f64p),
                                             _ptr(costs_off, _c_i64p)))
        else:
            _check(lib, lib.gh_viterbi(self.ctx.h, self.h, batch.h, _ptr(lat, _c_i32p), _ptr(end_cost, _c_f64p),
                                       _ptr(best_end, _c_i32p), _ptr(path, _c_i32p), _ptr(path_off, _c_i64p),
                                       _ptr(path_len, _c_i32p), _ptr(costs, _c_f64p), _ptr(costs_off, _c_i64p)))
        out = dict(best_end=best_end, end_off=end_off, end_cost_flat=end_cost,
                   end_cost=[end_cost[end_off[u]:end_off[u + 1]] for u in range(U)] if (U <= 2048 and want_end_cost) else None)
        if want_path and flat_paths:
            # the paths of all utterances back to back, no per-utterance arrays (2 000 of them cost 1.5 ms in train_words):
            # utterance u is path_flat[path_start[u] : path_start[u] + path_len[u]]
            n = path_len.astype(np.int64)
            pos = np.repeat(path_off[:-1] - (np.cumsum(n) - n), n) + np.arange(int(n.sum()))
            out["path_flat"], out["path_len"] = path[pos].astype(np.int64), n
            out["path_start"] = np.cumsum(n) - n
        elif want_path:
            out["paths"] = [path[path_off[u]:path_off[u] + path_len[u]].astype(np.int64) for u in range(U)]
        if want_costs:
            out["costs"] = [costs[costs_off[u]:costs_off[u + 1]].reshape(int(self.R[lidx[u]]), int(T[u]))
                            for u in range(U)]
        return out

    def viterbi_labels(self, batch, row_label, utt_lattice=None, max_labels=None, as_lists=True, want_end_cost=True):
        """A6 + A12 in one call: decode, keep the path on the device, return the decoded label sequences
        (main.py:59-67: first row of every emitting run between non-emitting rows).  row_label: one int32 array per
        graph (label per row, < 0 on non-emitting rows) or a single array when there is one graph.
        max_labels (scalar or [U]): upper bound on the labels per utterance; exceeding it raises BackendError.
        Returns dict(labels [list of int32 arrays], best_end [U], end_cost_flat, end_off, labels_flat, label_off,
        n_labels); as_lists=False leaves out the per-utterance list (at 10^5 utterances building it costs more host
        time than the decode takes on the GPU): utterance u is labels_flat[label_off[u] : label_off[u] + n_labels[u]].
        want_end_cost=False: the costs of the end rows stay on the device (best_end, the chosen end row, still comes back) --
        for a 10-word grammar that is 80 bytes per utterance of copy-back a label decode has no use for."""
        lib, U = self.ctx.lib, batch.U
        lat = None if utt_lattice is None else np.ascontiguousarray(utt_lattice, dtype=np.int32)
        lidx = np.zeros(U, dtype=np.int64) if lat is None else lat.astype(np.int64)
        if isinstance(row_label, np.ndarray) and row_label.ndim == 1 and self.L == 1:
            row_label = [row_label]
        rl = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.int32).reshape(-1) for r in row_label]))
        assert len(rl) == int(np.sum(self.R)), "row_label must give one label per graph row"
        T = batch.lengths
        end_off = end_cost = None
        if want_end_cost:
            n_end = np.asarray(self.n_end, dtype=np.int64)[lidx]
            end_off = np.concatenate([[0], np.cumsum(n_end)])
            end_cost = np.empty(int(end_off[-1]), dtype=np.float64)
        best_end = np.empty(U, dtype=np.int32)
        nlev = np.array([self.path_cap(l, 1) for l in range(self.L)], dtype=np.int64)[lidx]
        cap = np.where(T > 1, T * nlev // 2 + 1, 0)
        if max_labels is not None:   # the caller knows a tighter bound (e.g. K + 1 for a K-layer lattice): smaller copy-back
            cap = np.minimum(cap, np.asarray(max_labels, dtype=np.int64))
        n_labels = np.empty(U, dtype=np.int32)
        if not as_lists and U > 0:
            # packed: one device slot of the largest capacity per utterance, only the labels that exist come back
            mx = int(max(1, np.max(cap)))
            labels = np.empty(U * mx, dtype=np.int32)
            _check(lib, lib.gh_viterbi_labels_packed(self.ctx.h, self.h, batch.h, _ptr(lat, _c_i32p), _ptr(rl, _c_i32p), mx,
                                                     _ptr(end_cost, _c_f64p), _ptr(best_end, _c_i32p), _ptr(labels, _c_i32p),
                                                     labels.size, _ptr(n_labels, _c_i32p)))
            label_off = np.concatenate([[0], np.cumsum(n_labels)]).astype(np.int64)
            return dict(best_end=best_end, end_off=end_off, end_cost_flat=end_cost, labels_flat=labels[:label_off[-1]],
                        label_off=label_off[:-1], n_labels=n_labels)
        label_off = np.concatenate([[0], np.cumsum(cap)]).astype(np.int64)
        labels = np.empty(int(label_off[-1]), dtype=np.int32)
        _check(lib, lib.gh_viterbi_labels(self.ctx.h, self.h, batch.h, _ptr(lat, _c_i32p), _ptr(rl, _c_i32p),
                                          _ptr(end_cost, _c_f64p), _ptr(best_end, _c_i32p), _ptr(labels, _c_i32p),
                                          _ptr(label_off, _c_i64p), _ptr(n_labels, _c_i32p)))
        out = dict(best_end=best_end, end_off=end_off, end_cost_flat=end_cost, labels_flat=labels, label_off=label_off,
                   n_labels=n_labels)
        if as_lists:
            out["labels"] = [labels[label_off[u]:label_off[u] + n_labels[u]] for u in range(U)]
        return out

    SEGMENT_START = 1 << 30

    def align_segments(self, batch, utt_lattice=None, out=None):
        """Alignment + regrouping of continuous_train (continuous_speech.py:80-106) in one call, see gh_align_segments.
        Returns dict(frame_state int32 [N] (-1: the frame joins no state's data), segment_start bool [N], end_cost,
        best_end).  out: an int32 [N] buffer for frame_state that the caller keeps between calls (a fresh 5 MB numpy
        array is page-faulted in by the device-to-host copy: up to 30 ms against 3 ms for the whole call)."""
        lib, U = self.ctx.lib, batch.U
        lat = None if utt_lattice is None else np.ascontiguousarray(utt_lattice, dtype=np.int32)
        lidx = np.zeros(U, dtype=np.int64) if lat is None else lat.astype(np.int64)
        n_end = np.asarray(self.n_end, dtype=np.int64)[lidx]
        end_cost = np.empty(int(n_end.sum()))
        best_end = np.empty(U, dtype=np.int32)
        fs = out if (out is not None and out.dtype == np.int32 and out.shape == (batch.N,) and out.flags.c_contiguous) \
            else np.empty(batch.N, dtype=np.int32)
        _check(lib, lib.gh_align_segments(self.ctx.h, self.h, batch.h, _ptr(lat, _c_i32p), _ptr(end_cost, _c_f64p),
                                          _ptr(best_end, _c_i32p), _ptr(fs, _c_i32p)))
        start = (fs >= 0) & ((fs & self.SEGMENT_START) != 0)
        fs[start] &= ~self.SEGMENT_START
        return dict(frame_state=fs, segment_start=start, end_cost_flat=end_cost, best_end=best_end)

    def align_runs(self, batch, utt_lattice=None):
        """The alignment of `align_segments` with the RUNS as the result (gh_align_runs): dict(state int32 [R], start int64
        [R] (row of the batch), length int64 [R]) in utterance / time order -- ~N / 20 runs instead of N labels; end_cost,
        best_end as there."""
        lib, U = self.ctx.lib, batch.U
        lat = None if utt_lattice is None else np.ascontiguousarray(utt_lattice, dtype=np.int32)
        lidx = np.zeros(U, dtype=np.int64) if lat is None else lat.astype(np.int64)
        n_end = np.asarray(self.n_end, dtype=np.int64)[lidx]
        end_cost = np.empty(int(n_end.sum()))
        best_end = np.empty(U, dtype=np.int32)
        cap = int(np.max(self.R)) if len(np.atleast_1d(self.R)) else 1
        runs = np.empty((U, cap, 3), dtype=np.int32)
        cnt = np.empty(U, dtype=np.int32)
        _check(lib, lib.gh_align_runs(self.ctx.h, self.h, batch.h, _ptr(lat, _c_i32p), _ptr(end_cost, _c_f64p),
                                      _ptr(best_end, _c_i32p), cap, _ptr(runs, _c_i32p), _ptr(cnt, _c_i32p)))
        assert U == 0 or int(cnt.max()) <= cap, "align_runs: run table too small"
        keep = np.arange(cap)[None, :] < cnt[:, None]                     # [U, cap]: utterance / time order when flattened
        flat = runs[keep]
        start = flat[:, 1].astype(np.int64) + np.repeat(np.asarray(batch.offsets[:-1], dtype=np.int64), cnt)
        return dict(state=np.ascontiguousarray(flat[:, 0]), start=start, length=flat[:, 2].astype(np.int64),
                    end_cost_flat=end_cost, best_end=best_end)

    def forward_backward(self, batch, utt_lattice=None, want_matrices=False, want_occ=False, fetch_occ=True, want_self_xi=False):
        """A13: log P per utterance [+ log alpha / log beta / gamma matrices [R,T]] [+ occ [N,S]].
        want_occ keeps the frame x state occupancies resident in the batch (input of bw_accumulate);
        fetch_occ=False skips the [N,S] device-to-host copy."""
        lib, U = self.ctx.lib, batch.U
        lat = None if utt_lattice is None else np.ascontiguousarray(utt_lattice, dtype=np.int32)
        lidx = np.zeros(U, dtype=np.int64) if lat is None else lat.astype(np.int64)
        T = batch.lengths
        logp = np.empty(U)
        al = be = ga = off = occ = None
        if want_matrices:
            Rs = np.asarray(self.R, dtype=np.int64)[lidx]
            off = np.concatenate([[0], np.cumsum(Rs * T)]).astype(np.int64)
            al, be, ga = (np.empty(int(off[-1])) for _ in range(3))
        if want_occ and fetch_occ:
            occ = np.empty((batch.N, batch.S))
        xi = np.zeros(batch.S) if want_self_xi else None
        _check(lib, lib.gh_forward_backward(self.ctx.h, self.h, batch.h, _ptr(lat, _c_i32p), int(want_occ),
                                            _ptr(logp, _c_f64p), _ptr(al, _c_f64p), _ptr(be, _c_f64p),
                                            _ptr(ga, _c_f64p), _ptr(off, _c_i64p), _ptr(occ, _c_f64p), _ptr(xi, _c_f64p)))
        out = dict(logp=logp)
        if xi is not None:
            out["self_xi"] = xi   # expected number of self transitions per state
        if want_matrices:
            cut = lambda m: [m[off[u]:off[u + 1]].reshape(int(self.R[lidx[u]]), int(T[u])) for u in range(U)]
            out.update(alpha=cut(al), beta=cut(be), gamma=cut(ga))
        if occ is not None:
            out["occ"] = occ
        return out

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):  # a handle must not outlive its context (interpreter shutdown order)
                self.ctx.lib.gh_lattices_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def distance_matrix(ctx, x, y, var=None):
    """dist[i, n] between frames x[n] and templates y[i]; var None -> Euclid,
    [D] -> shared variance, [K,D] -> per-template variance (mahalanobis)."""
    x, y = _f64(x), _f64(y)
    K, D = y.shape
    out = np.empty((K, x.shape[0]))
    v = None if var is None else _f64(var).reshape(-1, D)
    _check(ctx.lib, ctx.lib.gh_distance_matrix(ctx.h, x.shape[0], K, D, _ptr(x, _c_f64p), _ptr(y, _c_f64p),
                                               _ptr(v, _c_f64p), 0 if v is None else v.shape[0],
                                               _ptr(out, _c_f64p)))
    return out
