"""ctypes binding of include/arkblst_amd.h.  Plumbing only — every computation happens in the HIP library."""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
SCALAR_CANONICAL, SCALAR_MONTGOMERY = 0, 1
G1_AFF, G1_JAC, G2_AFF, G2_JAC = 96, 144, 192, 288
FP12 = 576
MAX_WINDOWS = 37   # MI_MAX_WINDOWS


ABORT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)   # int (*check)(void *user): mi_msm_set_abort_check


class WindowInfo(C.Structure):
    _fields_ = [("window_bits", C.c_uint32), ("num_windows", C.c_uint32)]


class MsmError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        super().__init__(f"{where}: error {code} ({detail})")


class Profile(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("h2d_ms", "ingest_ms", "digits_ms", "scan_ms", "scatter_ms", "accumulate_ms",
                                          "reduce_ms", "combine_ms", "d2h_ms", "host_fold_ms", "total_ms")] + [
        ("window_bits", C.c_uint32), ("num_windows", C.c_uint32), ("n", C.c_uint64), ("accumulate_adds", C.c_uint64),
        ("work_items", C.c_uint32), ("max_items_per_bucket", C.c_uint32),
        ("window_groups", C.c_uint32), ("reserved", C.c_uint32),
        ("accumulate_clock_ghz", C.c_double), ("accumulate_clock_ticks", C.c_uint64), ("accumulate_ref_ticks", C.c_uint64)]


def profile_dict(p: Profile) -> dict:
    return {f: getattr(p, f) for f, _ in Profile._fields_}


class PairingProfile(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("h2d_ms", "lines_ms", "accumulate_ms", "miller_ms", "tree_ms", "host_ms", "total_ms")] + [
        ("n", C.c_uint64), ("pairs_per_accumulator", C.c_uint32), ("reserved", C.c_uint32)]


def lib_path(test_hooks: bool = False) -> str:
    """The product library, or the test build of the same sources (-DMI_TEST_HOOKS, csrc/test_hooks.h)."""
    return os.path.join(_HERE, "lib", "libarkblst_amd_test.so" if test_hooks else "libarkblst_amd.so")


_LIBS = {}


def load_library(test_hooks: bool = False):
    """Load the HIP library; raises (never falls back) when it has not been built."""
    if test_hooks not in _LIBS:
        path = lib_path(test_hooks)
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(path)
        vp, sz, u, i = C.c_void_p, C.c_size_t, C.c_uint, C.c_int
        L.mi_msm_init.argtypes = [C.POINTER(vp), C.POINTER(i), i]
        L.mi_msm_destroy.argtypes = [vp]
        L.mi_msm_destroy.restype = None
        L.mi_msm_num_devices.argtypes = [vp]
        for g in ("g1", "g2"):
            getattr(L, f"mi_msm_{g}_set_bases").argtypes = [vp, vp, sz]
            getattr(L, f"mi_msm_{g}_set_bases_precomputed").argtypes = [vp, vp, sz, u]
            getattr(L, f"mi_msm_{g}_set_bases_device").argtypes = [vp, vp, sz]
            getattr(L, f"mi_msm_{g}_set_bases_from_jacobian").argtypes = [vp, vp, sz]
            getattr(L, f"mi_msm_{g}_set_bases_from_compressed").argtypes = [vp, vp, sz, i, i, C.POINTER(sz)]
            getattr(L, f"mi_{g}_normalize_batch_device").argtypes = [vp, vp, sz, vp]
            getattr(L, f"mi_{g}_deserialize_batch_device").argtypes = [vp, vp, sz, i, i, vp, vp]
            getattr(L, f"mi_{g}_check_batch_device").argtypes = [vp, vp, sz, vp]
            getattr(L, f"mi_msm_{g}_validate_bases").argtypes = [vp, C.POINTER(sz)]
            getattr(L, f"mi_msm_{g}").argtypes = [vp, vp, vp, sz, u, vp]
            getattr(L, f"mi_msm_{g}_device").argtypes = [vp, vp, sz, u, vp]
            getattr(L, f"mi_{g}_sum").argtypes = [vp, sz, vp]
            if hasattr(L, f"mi_msm_{g}_device_windows"):   # absent from older builds that tools/ab_multi.sh swaps in for same-box A/Bs
                getattr(L, f"mi_msm_{g}_device_windows").argtypes = [vp, vp, sz, u, vp, C.POINTER(WindowInfo)]
                getattr(L, f"mi_{g}_fold_windows").argtypes = [vp, sz, sz, C.POINTER(WindowInfo), vp]
            getattr(L, f"mi_msm_{g}_batch").argtypes = [vp, C.POINTER(C.c_char_p), sz, sz, u, vp]
            getattr(L, f"mi_msm_{g}_batch_device").argtypes = [vp, C.POINTER(vp), sz, sz, u, vp]
            getattr(L, f"mi_{g}_normalize_batch").argtypes = [vp, vp, sz, vp]
        for g in ("g1", "g2"):
            getattr(L, f"mi_{g}_deserialize_batch").argtypes = [vp, vp, sz, i, i, vp, vp]
            getattr(L, f"mi_{g}_serialize_batch").argtypes = [vp, vp, sz, i, vp]
            getattr(L, f"mi_{g}_check_batch").argtypes = [vp, vp, sz, vp]
        L.mi_multi_miller_loop.argtypes = [vp, vp, vp, sz, vp]
        L.mi_multi_pairing.argtypes = [vp, vp, vp, sz, vp]
        L.mi_final_exponentiation.argtypes = [vp, vp]
        L.mi_msm_set_window_bits.argtypes = [vp, u]
        L.mi_msm_get_window_bits.argtypes = [vp, C.POINTER(u)]
        L.mi_msm_set_abort_check.argtypes = [vp, ABORT_FN, vp]
        L.mi_msm_set_pipeline.argtypes = [vp, C.POINTER(u), u]
        L.mi_msm_set_base_cache.argtypes = [vp, u]
        L.mi_msm_invalidate_base_cache.argtypes = [vp]
        L.mi_msm_base_cache_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(u)]
        L.mi_msm_device_id.argtypes = [vp, i]
        L.mi_msm_set_profile_level.argtypes = [vp, i]
        L.mi_msm_last_profile.argtypes = [vp, C.POINTER(Profile)]
        if hasattr(L, "mi_pairing_last_profile"):
            L.mi_pairing_last_profile.argtypes = [vp, C.POINTER(PairingProfile)]
        L.mi_msm_last_error.argtypes = [vp]
        L.mi_msm_last_error.restype = C.c_char_p
        L.mi_msm_strerror.argtypes = [i]
        L.mi_msm_strerror.restype = C.c_char_p
        if test_hooks:
            L.mi_test_fp_op.argtypes = [vp, i, vp, vp, vp, sz]
            L.mi_test_set_pairing.argtypes = [vp, u, u, i]
            L.mi_test_set_max_part.argtypes = [vp, sz]
            L.mi_test_fail_allocs.argtypes = [i]
            L.mi_test_fail_allocs.restype = None
            L.mi_test_set_no_peer.argtypes = [vp, i]
            L.mi_test_plan.argtypes = [sz, u, i, i, sz, C.POINTER(C.c_uint32)]
        _LIBS[test_hooks] = L
    return _LIBS[test_hooks]


def _buf(b):
    """bytes / bytearray / memoryview / numpy array -> (pointer, keepalive)."""
    if b is None:
        return None, None
    if isinstance(b, int):
        return C.c_void_p(b), None
    if isinstance(b, bytes):
        return C.cast(C.c_char_p(b), C.c_void_p), b
    mv = memoryview(b)
    arr = (C.c_char * mv.nbytes).from_buffer(mv) if not mv.readonly else (C.c_char * mv.nbytes).from_buffer_copy(mv)
    return C.cast(arr, C.c_void_p), arr


def check_runtime_order():
    """One HIP runtime per process.  The library links libamdhip64.so.7 by soname and a torch wheel bundles a runtime with
    the same soname: whichever loads first serves both.  If this library was loaded BEFORE torch, torch's bundled HSA runtime ends up next to
    the system one and torch.cuda reports "No HIP GPUs are available" — much later and far from the cause.  Refuse here, with the reason:
    torch is imported, but the libamdhip64 mapped into the process is not the one from torch's own lib directory."""
    torch = sys.modules.get("torch")
    if torch is None or os.environ.get("ARKBLST_AMD_SKIP_RUNTIME_ORDER_CHECK"):
        return
    tlib = os.path.join(os.path.dirname(getattr(torch, "__file__", "") or ""), "lib")
    if not os.path.exists(os.path.join(tlib, "libamdhip64.so")) and not any(f.startswith("libamdhip64.so") for f in (os.listdir(tlib) if os.path.isdir(tlib) else [])):
        return   # this torch does not bundle a HIP runtime: nothing to order
    try:
        mapped = {line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64.so" in line}
    except OSError:
        return
    foreign = sorted(m for m in mapped if not os.path.realpath(m).startswith(os.path.realpath(tlib)))
    if foreign:   # loaded in the right order there is exactly one runtime in the process: torch's
        raise ImportError("libarkblst_amd.so was loaded before torch: the process holds TWO HIP runtimes (" + ", ".join(sorted(mapped)) + "); torch's bundled "
                          "one under " + tlib + " then finds no devices ('No HIP GPUs are available').  Import torch BEFORE the first ark_blst_amd "
                          "context is created (INTEGRATION.md §6); ARKBLST_AMD_SKIP_RUNTIME_ORDER_CHECK=1 silences this check.")


class Context:
    """Owns one mi_ctx (streams, resident bases, scratch).  device_ids=None -> device 0 only.
    test_hooks=True loads the test build of the library (extra mi_test_* entry points; tests only)."""

    def __init__(self, device_ids=None, test_hooks: bool = False):
        self._L = load_library(test_hooks)
        check_runtime_order()
        self._h = C.c_void_p()
        if device_ids is None:
            device_ids = [0]
        ids = (C.c_int * len(device_ids))(*device_ids)
        rc = self._L.mi_msm_init(C.byref(self._h), ids, len(device_ids))
        if rc != 0:
            raise MsmError(rc, "mi_msm_init", self._L.mi_msm_strerror(rc).decode())

    def close(self):
        if self._h:
            self._L.mi_msm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, where):
        if rc != 0:
            raise MsmError(rc, where, (self._L.mi_msm_last_error(self._h) or b"").decode())

    def num_devices(self) -> int:
        return self._L.mi_msm_num_devices(self._h)

    def set_window_bits(self, c: int):
        self._check(self._L.mi_msm_set_window_bits(self._h, c), "mi_msm_set_window_bits")

    def set_abort_check(self, fn=None):
        """The reference driver's maybe_abort (src/gpu.rs:58,133-137): fn() -> bool is asked at the start of every MSM call and between the
        passes of a long one; True ends the call with MI_E_ABORTED (-8).  None removes the check."""
        self._abort_cb = ABORT_FN(lambda _user: 1 if fn() else 0) if fn else C.cast(None, ABORT_FN)   # kept alive by the context
        self._check(self._L.mi_msm_set_abort_check(self._h, self._abort_cb, None), "mi_msm_set_abort_check")

    def get_window_bits(self) -> int:
        v = C.c_uint(0)
        self._check(self._L.mi_msm_get_window_bits(self._h, C.byref(v)), "mi_msm_get_window_bits")
        return v.value

    def set_pipeline(self, weights=None):
        """Window groups of a pipelined call (mi_msm_set_pipeline): None = built-in choice, [1] = off, else relative sizes, top windows first."""
        w = list(weights or [])
        arr = (C.c_uint * max(1, len(w)))(*w)
        self._check(self._L.mi_msm_set_pipeline(self._h, arr, len(w)), "mi_msm_set_pipeline")

    def set_base_cache(self, entries: int):
        """base-set cache of the stateless call shape (mi_msm_set_base_cache): 0 = off"""
        self._check(self._L.mi_msm_set_base_cache(self._h, entries), "mi_msm_set_base_cache")

    def invalidate_base_cache(self):
        self._check(self._L.mi_msm_invalidate_base_cache(self._h), "mi_msm_invalidate_base_cache")

    def base_cache_stats(self) -> dict:
        h, m, e = C.c_uint64(0), C.c_uint64(0), C.c_uint(0)
        self._check(self._L.mi_msm_base_cache_stats(self._h, C.byref(h), C.byref(m), C.byref(e)), "mi_msm_base_cache_stats")
        return {"hits": h.value, "misses": m.value, "entries": e.value}

    def device_id(self, slot: int = 0) -> int:
        return self._L.mi_msm_device_id(self._h, slot)

    def set_profile_level(self, level: int):
        """0: no timing events, 1 (default): the accumulate kernel's interval only, 2: every phase of mi_profile."""
        self._check(self._L.mi_msm_set_profile_level(self._h, level), "mi_msm_set_profile_level")

    def set_bases(self, group: str, bases, n: int):
        p, keep = _buf(bases)
        self._check(getattr(self._L, f"mi_msm_{group}_set_bases")(self._h, p, n), f"mi_msm_{group}_set_bases")

    def set_bases_precomputed(self, group: str, bases, n: int, window_bits: int = 0):
        """Resident SRS with 2^(c j) P tables: all windows of later MSMs share one bucket set (opt-in, W x the memory)."""
        p, keep = _buf(bases)
        self._check(getattr(self._L, f"mi_msm_{group}_set_bases_precomputed")(self._h, p, n, window_bits),
                    f"mi_msm_{group}_set_bases_precomputed")

    def set_bases_device(self, group: str, d_bases_ptr: int, n: int):
        """Resident base set from affine points (reference form) already in device memory."""
        self._check(getattr(self._L, f"mi_msm_{group}_set_bases_device")(self._h, C.c_void_p(d_bases_ptr), n), f"mi_msm_{group}_set_bases_device")

    def set_bases_from_jacobian(self, group: str, jac, n: int):
        """Resident base set from host Jacobian points: normalize_batch on the GPU on the way in."""
        p, keep = _buf(jac)
        self._check(getattr(self._L, f"mi_msm_{group}_set_bases_from_jacobian")(self._h, p, n), f"mi_msm_{group}_set_bases_from_jacobian")

    def set_bases_from_compressed(self, group: str, data, n: int, compressed: bool = True, validate: bool = True) -> int:
        """Resident base set from serialized points, decoded (and checked) on the GPU.  Returns the number of rejected encodings:
        0 = installed; otherwise nothing was installed (the call's MI_E_INVALID is swallowed, every other error raises)."""
        p, keep = _buf(data)
        rej = C.c_size_t(0)
        rc = getattr(self._L, f"mi_msm_{group}_set_bases_from_compressed")(self._h, p, n, 1 if compressed else 0, 1 if validate else 0, C.byref(rej))
        if rc != 0 and not (rc == -1 and rej.value):
            self._check(rc, f"mi_msm_{group}_set_bases_from_compressed")
        return rej.value

    def normalize_batch_device(self, group: str, d_in_ptr: int, n: int, d_out_ptr: int):
        self._check(getattr(self._L, f"mi_{group}_normalize_batch_device")(self._h, C.c_void_p(d_in_ptr), n, C.c_void_p(d_out_ptr)), f"mi_{group}_normalize_batch_device")

    def deserialize_batch_device(self, group: str, d_bytes_ptr: int, n: int, compressed: bool, validate: bool, d_out_ptr: int, d_status_ptr: int):
        self._check(getattr(self._L, f"mi_{group}_deserialize_batch_device")(self._h, C.c_void_p(d_bytes_ptr), n, 1 if compressed else 0,
                                                                              1 if validate else 0, C.c_void_p(d_out_ptr), C.c_void_p(d_status_ptr)),
                    f"mi_{group}_deserialize_batch_device")

    def check_batch_device(self, group: str, d_points_ptr: int, n: int, d_status_ptr: int):
        self._check(getattr(self._L, f"mi_{group}_check_batch_device")(self._h, C.c_void_p(d_points_ptr), n, C.c_void_p(d_status_ptr)), f"mi_{group}_check_batch_device")

    def validate_bases(self, group: str) -> int:
        """Valid::check of the resident base set on the GPU; returns the number of points that fail (0 = the set is recorded as valid)"""
        bad = C.c_size_t(0)
        self._check(getattr(self._L, f"mi_msm_{group}_validate_bases")(self._h, C.byref(bad)), "mi_msm_validate_bases")
        return bad.value

    def msm(self, group: str, bases, scalars, n: int, scalar_fmt: int = SCALAR_CANONICAL) -> bytes:
        """bases=None uses the resident set. Returns the Jacobian result bytes (blst_p1 / blst_p2)."""
        out = C.create_string_buffer(G1_JAC if group == "g1" else G2_JAC)
        pb, kb = _buf(bases)
        ps, ks = _buf(scalars)
        self._check(getattr(self._L, f"mi_msm_{group}")(self._h, pb, ps, n, scalar_fmt, out), f"mi_msm_{group}")
        return out.raw

    def msm_batch(self, group: str, scalar_vectors, n: int, scalar_fmt: int = SCALAR_CANONICAL):
        """k MSMs over the resident base set in one call (two in flight on the context's lanes) -> list of Jacobian bytes."""
        k = len(scalar_vectors)
        size = G1_JAC if group == "g1" else G2_JAC
        out = C.create_string_buffer(size * max(k, 1))
        arr = (C.c_char_p * max(k, 1))(*[bytes(v) for v in scalar_vectors])
        self._check(getattr(self._L, f"mi_msm_{group}_batch")(self._h, arr, k, n, scalar_fmt, out), f"mi_msm_{group}_batch")
        return [out.raw[size * j:size * (j + 1)] for j in range(k)]

    def msm_batch_device(self, group: str, d_scalar_ptrs, n: int, scalar_fmt: int = SCALAR_CANONICAL):
        """msm_batch with the scalar vectors already in device memory (a list of device pointers)."""
        k = len(d_scalar_ptrs)
        size = G1_JAC if group == "g1" else G2_JAC
        out = C.create_string_buffer(size * max(k, 1))
        arr = (C.c_void_p * max(k, 1))(*[C.c_void_p(p) for p in d_scalar_ptrs])
        self._check(getattr(self._L, f"mi_msm_{group}_batch_device")(self._h, arr, k, n, scalar_fmt, out), f"mi_msm_{group}_batch_device")
        return [out.raw[size * j:size * (j + 1)] for j in range(k)]

    def msm_device(self, group: str, d_scalars_ptr: int, n: int, scalar_fmt: int = SCALAR_CANONICAL) -> bytes:
        out = C.create_string_buffer(G1_JAC if group == "g1" else G2_JAC)
        self._check(getattr(self._L, f"mi_msm_{group}_device")(self._h, C.c_void_p(d_scalars_ptr), n, scalar_fmt, out),
                    f"mi_msm_{group}_device")
        return out.raw

    def msm_device_windows(self, group: str, d_scalars_ptr: int, n: int, scalar_fmt: int, d_out_ptr: int):
        """The pipeline of msm_device up to the per-window sums, left in the caller's DEVICE buffer (room for MAX_WINDOWS points):
        the exchange step of a one-process-per-GPU deployment starts from device memory.  Returns (window_bits, num_windows)."""
        info = WindowInfo()
        self._check(getattr(self._L, f"mi_msm_{group}_device_windows")(self._h, C.c_void_p(d_scalars_ptr), n, scalar_fmt,
                                                                        C.c_void_p(d_out_ptr), C.byref(info)), f"mi_msm_{group}_device_windows")
        return int(info.window_bits), int(info.num_windows)

    def normalize_batch(self, group: str, jac: bytes, into=None):
        """CurveGroup::normalize_batch: packed Jacobian points -> packed affine points (infinity -> zeros).  into: a caller-owned writable
        buffer (numpy uint8 array) to fill instead of a fresh bytes object."""
        jb, ab = (G1_JAC, G1_AFF) if group == "g1" else (G2_JAC, G2_AFF)
        n = len(jac) // jb
        if into is not None:
            po, ko = _buf(into)
            self._check(getattr(self._L, f"mi_{group}_normalize_batch")(self._h, jac, n, po), f"mi_{group}_normalize_batch")
            return into
        out = C.create_string_buffer(ab * n)
        self._check(getattr(self._L, f"mi_{group}_normalize_batch")(self._h, jac, n, out), f"mi_{group}_normalize_batch")
        return out.raw

    def deserialize_batch(self, group: str, data: bytes, compressed: bool = True, validate: bool = True, into=None):
        """CanonicalDeserialize for many points: returns (packed blst affine points, status bytes)."""
        unit = 48 if group == "g1" else 96
        size = unit if compressed else 2 * unit
        n = len(data) // size
        if into is not None:   # caller-owned, writable, already-touched buffers (numpy uint8 arrays): what a caller that reuses its vectors pays
            po, ko = _buf(into[0])
            ps, ks = _buf(into[1])
            self._check(getattr(self._L, f"mi_{group}_deserialize_batch")(self._h, data, n, int(compressed), int(validate), po, ps),
                        f"mi_{group}_deserialize_batch")
            return into
        out = C.create_string_buffer(2 * unit * n)
        st = C.create_string_buffer(n)
        self._check(getattr(self._L, f"mi_{group}_deserialize_batch")(self._h, data, n, int(compressed), int(validate), out, st),
                    f"mi_{group}_deserialize_batch")
        return out.raw, st.raw

    def check_batch(self, group: str, points: bytes) -> bytes:
        """Valid::batch_check over affine points: status bytes (0 valid, 2 not on the curve, 3 not in the subgroup)"""
        aff = G1_AFF if group == "g1" else G2_AFF
        n = len(points) // aff
        st = C.create_string_buffer(n)
        p, keep = _buf(points)
        self._check(getattr(self._L, f"mi_{group}_check_batch")(self._h, p, n, st), f"mi_{group}_check_batch")
        return st.raw

    def serialize_batch(self, group: str, points: bytes, compressed: bool = True) -> bytes:
        unit = 48 if group == "g1" else 96
        n = len(points) // (2 * unit)
        out = C.create_string_buffer((unit if compressed else 2 * unit) * n)
        self._check(getattr(self._L, f"mi_{group}_serialize_batch")(self._h, points, n, int(compressed), out), f"mi_{group}_serialize_batch")
        return out.raw

    def g1_deserialize_batch(self, data: bytes, compressed: bool = True, validate: bool = True):
        return self.deserialize_batch("g1", data, compressed, validate)

    def g1_serialize_batch(self, points: bytes, compressed: bool = True) -> bytes:
        return self.serialize_batch("g1", points, compressed)

    def multi_miller_loop(self, g1_affine: bytes, g2_affine: bytes) -> bytes:
        """Pairing::multi_miller_loop over packed blst_p1_affine / blst_p2_affine arrays -> blst_fp12 (576 B)."""
        n = len(g1_affine) // G1_AFF
        assert len(g1_affine) == n * G1_AFF and len(g2_affine) == n * G2_AFF, "one G2 point per G1 point"
        out = C.create_string_buffer(FP12)
        self._check(self._L.mi_multi_miller_loop(self._h, g1_affine, g2_affine, n, out), "mi_multi_miller_loop")
        return out.raw

    def multi_pairing(self, g1_affine: bytes, g2_affine: bytes) -> bytes:
        """prod_i e(P_i, Q_i): Miller loops on the GPU, multiplication tree, final exponentiation."""
        n = len(g1_affine) // G1_AFF
        assert len(g1_affine) == n * G1_AFF and len(g2_affine) == n * G2_AFF, "one G2 point per G1 point"
        out = C.create_string_buffer(FP12)
        self._check(self._L.mi_multi_pairing(self._h, g1_affine, g2_affine, n, out), "mi_multi_pairing")
        return out.raw

    def profile(self) -> dict:
        return profile_dict(self.profile_raw())

    def profile_raw(self) -> Profile:
        """The last call's profile as the C struct (a few microseconds: for timed loops; profile_dict() turns it into a dict later)."""
        p = Profile()
        self._check(self._L.mi_msm_last_profile(self._h, C.byref(p)), "mi_msm_last_profile")
        return p

    def pairing_profile(self) -> dict:
        p = PairingProfile()
        self._check(self._L.mi_pairing_last_profile(self._h, C.byref(p)), "mi_pairing_last_profile")
        return {f: getattr(p, f) for f, _ in PairingProfile._fields_ if f != "reserved"}

    # ---- test build only (Context(..., test_hooks=True))
    def test_fp_op(self, op: int, a: bytes, b: bytes) -> bytes:
        n = len(a) // 48
        out = C.create_string_buffer(48 * n)
        self._check(self._L.mi_test_fp_op(self._h, op, a, b, out, n), "mi_test_fp_op")
        return out.raw

    def test_set_pairing(self, share: int = 0, batch: int = 0, single_lane: bool = False):
        self._check(self._L.mi_test_set_pairing(self._h, share, batch, int(single_lane)), "mi_test_set_pairing")

    def test_set_max_part(self, points: int):
        self._check(self._L.mi_test_set_max_part(self._h, points), "mi_test_set_max_part")

    def test_fail_allocs(self, count: int):
        self._L.mi_test_fail_allocs(count)

    def test_set_no_peer(self, no_peer: bool):
        self._check(self._L.mi_test_set_no_peer(self._h, int(no_peer)), "mi_test_set_no_peer")


def test_plan(n: int, forced_c: int = 0, group: str = "g1", shared: bool = False, stride: int = 0, fold: bool = False) -> dict:
    """The window-size plan of an n-point call (test build; host only, needs no device).  fold: the plan of a validated resident set."""
    out = (C.c_uint32 * 13)()
    rc = load_library(True).mi_test_plan(n, forced_c, 0 if group == "g1" else 1, int(shared) | (2 if fold else 0), stride, out)
    if rc != 0:
        raise MsmError(rc, "mi_test_plan")
    keys = ("c", "nwin", "bwin", "coop_L", "chunk_buckets", "logT", "lo_bits", "serial", "chunks_per_win")
    d = {k: int(out[i]) for i, k in enumerate(keys)}
    d["nbuckets"] = (int(out[9]) << 32) | int(out[10])
    d["nchunks"] = int(out[11])
    d["serial_L"] = int(out[12])
    return d


def final_exponentiation(f: bytes) -> bytes:
    """Pairing::final_exponentiation on one blst_fp12 (host tail; needs no device)."""
    assert len(f) == FP12
    out = C.create_string_buffer(FP12)
    rc = load_library().mi_final_exponentiation(f, out)
    if rc != 0:
        raise MsmError(rc, "mi_final_exponentiation")
    return out.raw


def _sum(group: str, partials) -> bytes:
    L = load_library()
    size = G1_JAC if group == "g1" else G2_JAC
    blob = b"".join(partials)
    assert len(blob) % size == 0
    out = C.create_string_buffer(size)
    rc = getattr(L, f"mi_{group}_sum")(blob, len(blob) // size, out)
    if rc != 0:
        raise MsmError(rc, f"mi_{group}_sum")
    return out.raw


def fold_windows(group: str, windows, n_ranks: int, rank_stride: int, window_bits: int, num_windows: int) -> bytes:
    """Host fold of gathered per-window sums (mi_g{1,2}_fold_windows): windows[r * rank_stride + w], ranks added in index order,
    Horner over the windows.  `windows`: bytes-like of n_ranks * rank_stride Jacobian points."""
    size = G1_JAC if group == "g1" else G2_JAC
    p, keep = _buf(windows)
    assert memoryview(windows).nbytes >= n_ranks * rank_stride * size or n_ranks * num_windows == 0
    info = WindowInfo(window_bits, num_windows)
    out = C.create_string_buffer(size)
    rc = getattr(load_library(), f"mi_{group}_fold_windows")(p, n_ranks, rank_stride, C.byref(info), out)
    if rc != 0:
        raise MsmError(rc, f"mi_{group}_fold_windows")
    return out.raw


def g1_sum(partials) -> bytes:
    return _sum("g1", partials)


def g2_sum(partials) -> bytes:
    return _sum("g2", partials)


# ---------------------------------------------------------------------------------------------- multi-process exchange
class RcclTiming(C.Structure):
    _fields_ = [("msm_ms", C.c_double), ("exchange_ms", C.c_double), ("window_bits", C.c_uint32), ("num_windows", C.c_uint32),
                ("repeats", C.c_uint32), ("bytes_per_rank", C.c_uint32)]


RCCL_UNIQUE_ID_BYTES = 128
_RCCL = []


def rccl_lib_path() -> str:
    return os.path.join(_HERE, "lib", "libarkblst_amd_rccl.so")


def load_rccl_library():
    """libarkblst_amd_rccl.so (include/arkblst_amd_rccl.h): the exchange step of a one-process-per-GPU deployment.  Loaded on demand:
    it pulls in librccl.so.1, which a single-GPU user never needs."""
    if not _RCCL:
        load_library(False)   # the product library first: the exchange library's NEEDED entry resolves to the same object
        path = rccl_lib_path()
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        vp, sz, u, i = C.c_void_p, C.c_size_t, C.c_uint, C.c_int
        L.mi_rccl_get_unique_id.argtypes = [vp]
        L.mi_rccl_comm_create.argtypes = [C.POINTER(vp), vp, vp, i, i]
        L.mi_rccl_comm_attach.argtypes = [C.POINTER(vp), vp, vp]
        L.mi_rccl_comm_set_timeout_ms.argtypes = [vp, C.c_double]
        L.mi_rccl_comm_destroy.argtypes = [vp]
        L.mi_rccl_comm_destroy.restype = None
        L.mi_rccl_comm_size.argtypes = [vp]
        L.mi_rccl_comm_rank.argtypes = [vp]
        for g in ("g1", "g2"):
            getattr(L, f"mi_msm_{g}_allgather_fold").argtypes = [vp, vp, sz, u, vp]
        L.mi_rccl_last_timing.argtypes = [vp, C.POINTER(RcclTiming)]
        L.mi_rccl_last_error.restype = C.c_char_p
        _RCCL.append(L)
    return _RCCL[0]


def rccl_unique_id() -> bytes:
    """ncclGetUniqueId through the library: ONE rank calls it and hands the 128 bytes to the others."""
    L = load_rccl_library()
    buf = C.create_string_buffer(RCCL_UNIQUE_ID_BYTES)
    rc = L.mi_rccl_get_unique_id(buf)
    if rc != 0:
        raise MsmError(rc, "mi_rccl_get_unique_id", (L.mi_rccl_last_error() or b"").decode())
    return buf.raw


class RcclComm:
    """One rank's end of the exchange (mi_rccl_comm): collective construction, collective allgather_fold."""

    def __init__(self, ctx: Context, unique_id: bytes, n_ranks: int, rank: int):
        self._L = load_rccl_library()
        self._h = C.c_void_p()
        self._ctx = ctx   # the context must outlive the communicator
        rc = self._L.mi_rccl_comm_create(C.byref(self._h), ctx._h, unique_id, n_ranks, rank)
        if rc != 0:
            raise MsmError(rc, "mi_rccl_comm_create", (self._L.mi_rccl_last_error() or b"").decode())

    def close(self):
        if self._h:
            self._L.mi_rccl_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_timeout_ms(self, ms: float):
        """How long a rank waits inside the exchange for its peers (default 60 s; 0 = for ever); on expiry the communicator is aborted (MI_E_COMM)."""
        rc = self._L.mi_rccl_comm_set_timeout_ms(self._h, float(ms))
        if rc != 0:
            raise MsmError(rc, "mi_rccl_comm_set_timeout_ms", (self._L.mi_rccl_last_error() or b"").decode())

    def size(self) -> int:
        return self._L.mi_rccl_comm_size(self._h)

    def rank(self) -> int:
        return self._L.mi_rccl_comm_rank(self._h)

    def allgather_fold(self, group: str, d_scalars_ptr: int, n: int, scalar_fmt: int = SCALAR_CANONICAL) -> bytes:
        out = C.create_string_buffer(G1_JAC if group == "g1" else G2_JAC)
        rc = getattr(self._L, f"mi_msm_{group}_allgather_fold")(self._h, C.c_void_p(d_scalars_ptr), n, scalar_fmt, out)
        if rc != 0:
            raise MsmError(rc, f"mi_msm_{group}_allgather_fold", (self._L.mi_rccl_last_error() or b"").decode())
        return out.raw

    def timing_raw(self) -> RcclTiming:
        t = RcclTiming()
        self._L.mi_rccl_last_timing(self._h, C.byref(t))
        return t

    def timing(self) -> dict:
        t = self.timing_raw()
        return {f: getattr(t, f) for f, _ in RcclTiming._fields_}
