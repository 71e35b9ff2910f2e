"""ctypes binding of libpk_amd.so (include/pk_amd.h).

The library is the product: if it cannot be loaded the engine raises - there is
no CPU or eager-torch fallback anywhere in this package.  ctypes releases the
GIL around every call, so PyTorch-Kaldi's chunk-prefetch thread (core.py:25-34,
510-512) keeps running while kernels are enqueued.
"""
import ctypes
import os
import re
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libpk_amd.so")
HEADER = os.path.join(HERE, "..", "include", "pk_amd.h")

_lock = threading.Lock()
_lib = None

c_void_p, c_int, c_float, c_double, c_int64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_int64
P = c_void_p  # every device pointer crosses the ABI as a plain address

# name -> (restype, argtypes); mirrors include/pk_amd.h one to one
SIGNATURES = {
    "pk_version": (c_int, []),
    "pk_last_error": (ctypes.c_char_p, []),
    "pk_num_cu": (c_int, []),
    "pk_gemm": (c_int, [P, c_int, c_int, c_int, c_int, c_float, P, c_int64, c_int64, P, c_int64, c_int64, c_float, P,
                        c_int64, P, c_int, P]),
    "pk_gemm_f32_set_form": (None, [c_int]),
    "pk_gemm_bf16_tile_m": (c_int, [c_int]),
    "pk_gemm_bf16_auto_splitk": (c_int, [c_int, c_int, c_int]),
    "pk_gemm_bf16_set_tile": (None, [c_int]),
    "pk_gemm_bf16": (c_int, [P, c_int, c_int, c_int, c_float, P, c_int64, c_int, P, c_int64, c_int, c_float, P, c_int64, P,
                             c_int, P]),
    "pk_gemm_bf16_stats_floats": (c_int64, [c_int, c_int]),
    "pk_gemm_bf16_stats": (c_int, [P, c_int, c_int, c_int, c_float, P, c_int64, c_int, P, c_int64, c_int, P, c_int64, P, P, P]),
    "pk_bn_stats_merge": (c_int, [P, P, c_int, c_int64, P, P]),
    "pk_cvt_bf16": (c_int, [P, P, c_int64, c_int64, c_int, c_int, c_int, P, c_int64]),
    "pk_bn_partial_floats": (c_int64, [c_int64, c_int64]),
    "pk_bn_stats": (c_int, [P, P, c_int64, c_int64, c_int64, P, P, P]),
    "pk_bn_finalize": (c_int, [P, c_int64, P, P, P, P, c_float, P, P, P, P, c_float, c_double]),
    "pk_bn_finalize_gates": (c_int, [P, c_int, c_int, P, P, P, P, c_float, P, P, P, P, P, c_float, c_double]),
    "pk_bn_stats_merge_finalize_gates": (c_int, [P, P, c_int, c_int, c_int, P, P, P, P, c_float, P, P, P, P, P, c_float,
                                                 c_double]),
    "pk_affine_act_fwd": (c_int, [P, P, c_int64, c_int64, c_int64, P, P, c_int, P, P, c_int64]),
    "pk_act_bwd": (c_int, [P, P, P, P, c_int, c_int64, P]),
    "pk_bn_bwd_reduce": (c_int, [P, P, P, c_int64, P, c_int64, c_int64, c_int64, P, P, c_float, P, P, P]),
    "pk_bn_bwd_apply": (c_int, [P, P, P, c_int64, P, c_int64, c_int64, c_int64, P, P, c_float, P, P, P, c_double, P,
                                c_int64]),
    "pk_bn_bwd_bf16": (c_int, [P, P, P, c_int64, c_int, c_int, P, c_int64, c_int64, P, P, c_float, P, c_double, P, P, P, P,
                               c_int64, P, P]),
    "pk_colsum": (c_int, [P, P, P, c_int64, c_int64, c_int64, P, P]),
    "pk_add": (c_int, [P, P, P, c_int64, P]),
    "pk_mt19937_bernoulli": (c_int, [P, P, c_int64, c_float, P]),
    "pk_sinc_bank_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_float, c_float, c_float, P, P, P]),
    "pk_sinc_bank_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_float, c_float, c_float, P, P]),
    "pk_ln_last_act_drop_fwd": (c_int, [P, P, c_int64, c_int, c_int, P, P, c_float, c_int, P, P, P, P, P]),
    "pk_ln_last_act_drop_bwd": (c_int, [P, P, P, P, P, c_int64, c_int, c_int, P, P, P, c_float, c_int, P, P]),
    "pk_layernorm_fwd": (c_int, [P, P, c_int64, c_int64, P, P, c_float, P, P, P]),
    "pk_layernorm_bwd": (c_int, [P, P, P, c_int64, c_int64, P, P, P, c_float, P, P]),
    "pk_logsoftmax_fwd": (c_int, [P, P, c_int64, c_int64, P]),
    "pk_logsoftmax_bwd": (c_int, [P, P, P, c_int64, c_int64, P]),
    "pk_logsoftmax_bwd_ld": (c_int, [P, P, P, c_int64, c_int64, P, c_int64]),
    "pk_logsoftmax_fwd_ld": (c_int, [P, P, c_int64, c_int64, c_int64, P]),
    "pk_logsoftmax_fwd_ld_argmax": (c_int, [P, P, c_int64, c_int64, c_int64, P, P]),
    "pk_logsoftmax_bwd_bf16_partial_floats": (c_int64, [c_int64, c_int64]),
    "pk_logsoftmax_bwd_bf16": (c_int, [P, P, P, c_int64, c_int64, P, c_int64, P, P]),
    "pk_nll_err_partial_floats": (c_int64, [c_int64]),
    "pk_nll_err_fwd": (c_int, [P, P, P, c_int64, c_int64, c_int64, P, P, P, P]),
    "pk_nll_err_fwd_argmax": (c_int, [P, P, P, P, c_int64, c_int64, c_int64, P, P, P, P]),
    "pk_nll_logsoftmax_bwd_bf16": (c_int, [P, P, P, P, P, c_int64, c_int64, c_int64, P, c_int64, P, P]),
    "pk_logsoftmax_bwd_bf16_p": (c_int, [P, P, P, c_int64, c_int64, P, c_int64, c_int64, P, P]),
    "pk_nll_logsoftmax_bwd_bf16_p": (c_int, [P, P, P, P, P, c_int64, c_int64, c_int64, P, c_int64, c_int64, P, P]),
    "pk_rec_num_saved": (c_int, [c_int]),
    "pk_rec_num_gates": (c_int, [c_int]),
    "pk_conv1d_pool_dgrad": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "pk_conv_bf16_covers": (c_int, [c_int, c_int, c_int, c_int]),
    "pk_conv_bf16_work_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "pk_conv1d_pool_fwd_bf16": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "pk_conv1d_pool_bwd_bf16": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "pk_bn_act_bwd_small_covers": (c_int, [c_int64, c_int64]),
    "pk_bn_act_bwd_small": (c_int, [P, P, P, P, c_int, P, P, P, c_float, P, c_int64, c_int64, P, c_int64, P, P, P, P, P, P, P]),
    "pk_rec_work_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "pk_rec_ln_saved_floats": (c_int64, [c_int, c_int, c_int, c_int]),
    "pk_rec_ln_work_floats": (c_int64, [c_int, c_int, c_int, c_int]),
    "pk_rec_fwd": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, c_float, P, P, P,
                           P, P, P]),
    "pk_rec_bwd": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_float, P, P, P, P, P, P,
                           P, P, P, P]),
    "pk_rec_self_fill": (c_int, [c_int]),
    "pk_rec_plan_cus": (c_int, [c_int, c_int]),
    "pk_gemm_bf16_auto_splitk_cus": (c_int, [c_int, c_int, c_int, c_int]),
    "pk_rec_fwd_bf16": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, c_float, P, P, P, c_int64, c_int]),
    "pk_rec_bwd_bf16": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_float, P, P, P, P, P, c_int64, c_int]),
    "pk_rec2p_fwd_bf16_ln": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, c_float, P, P, c_float, P, P, P,
                                     P, P, c_int64, c_int, P]),
    "pk_rec2p_bwd_bf16_ln": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_float, P, c_float, P, P, P, P, P,
                                     c_int64, c_int, P, P, P]),
    "pk_rec_fwd_bf16_ln": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, c_float, P, P, c_float, P, P, P,
                                   P, c_int64, c_int, P]),
    "pk_rec_bwd_bf16_ln": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_float, P, c_float, P, P, P, P, P, P,
                                   c_int64, c_int, P, P, P]),
    "pk_rec2p_fwd_bf16": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, c_float, P, P, P, P, c_int64, c_int]),
    "pk_rec2p_bwd_bf16": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_float, P, P, P, P, c_int64, c_int]),
    "pk_persist2_set_trace": (None, [P]),
    "pk_persist2_set_empty_step": (None, [c_int]),
    "pk_persist2_set_mode": (None, [c_int]),
    "pk_persist2_set_poll_delay": (None, [c_int]),
    "pk_persist2_set_lstm_waves": (None, [c_int]),
    "pk_rec_helper_set_mode": (None, [c_int]),
    "pk_rec_helper_get_mode": (c_int, []),
    "pk_persist2_get_lstm_waves": (c_int, []),
    "pk_persist2_error_count": (ctypes.c_uint, []),
    "pk_persist2_error_reset": (None, []),
    "pk_conv1d_pool_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "pk_conv_fwd_work_floats": (c_int64, [c_int, c_int, c_int]),
    "pk_conv_partial_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "pk_conv1d_pool_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P]),
    "pk_ark_open": (P, [ctypes.c_char_p, c_int64]),
    "pk_ark_close": (None, [P]),
    "pk_ark_next": (c_int, [P, c_int, ctypes.c_char_p, c_int, ctypes.POINTER(c_int64), ctypes.POINTER(c_int64)]),
    "pk_ark_read": (c_int, [P, P]),
    "pk_ark_skip": (c_int, [P]),
    "pk_ivec_next": (c_int, [P, ctypes.c_char_p, c_int, ctypes.POINTER(c_int64)]),
    "pk_ivec_read": (c_int, [P, P]),
    "pk_context_window": (c_int, [P, c_int64, c_int64, c_int, c_int, P]),
    "pk_mean_var_norm": (c_int, [P, c_int64, c_int64]),
    "pk_rmsprop_step": (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float]),
    "pk_sgd_step": (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_int]),
    "pk_adam_step": (c_int, [P, P, P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int]),
    "pk_fused_step": (c_int, [P, c_int, P, P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_int, P, c_int, P]),
    "pk_persist_error_count": (ctypes.c_uint, []),
    "pk_persist_error_reset": (None, []),
    "pk_linear_bn_act_bf16_covers": (c_int, [c_int64, c_int64, c_int64]),
    "pk_linear_bn_act_bf16": (c_int, [P, c_int, c_int, c_int, P, c_int64, P, c_int64, P, P, P, c_float, c_float, P, P, c_int, P,
                                      P, P, P, P, c_int64, P, P]),
    "pk_gemm_bf16_small_splitk": (c_int, [c_int, c_int, c_int]),
    "pk_linear_bn_act_bf16_sk": (c_int, [P, c_int, c_int, c_int, P, c_int64, P, c_int64, P, P, P, c_float, c_float, P, P, c_int, P,
                                         P, P, P, P, c_int64, P, P, c_int, P]),
    "pk_selftest_mfma": (c_int, [P, ctypes.POINTER(c_int)]),
    "pk_selftest_permlane": (c_int, [P, ctypes.POINTER(c_int)]),
    "pk_selftest_dpp_row_sum": (c_int, [P, ctypes.POINTER(c_int)]),
    "pk_selftest_cu_hog": (c_int, [P, c_int, c_int, P]),
}


def declared_symbols():
    """Every function name declared in include/pk_amd.h (used by the ABI test)."""
    with open(HEADER) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(pk_[a-z0-9_]+)\s*\(", text)))


class PkError(RuntimeError):
    pass


def _map_torch_hip_runtime():
    """One HIP runtime per process, by construction instead of by import order: libpk_amd.so NEEDs
    `libamdhip64.so.7`; torch ships its own runtime as `torch/lib/libamdhip64.so` - a FILE name the loader never searches
    for, whatever RUNPATH says - whose SONAME is that very `libamdhip64.so.7`.  Mapping torch's copy explicitly first makes
    the loader resolve our NEEDED entry to it (an already-loaded object with a matching SONAME wins over every search
    path), whether or not `import torch` has initialised anything yet."""
    import torch

    hip = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip):
        ctypes.CDLL(hip, mode=ctypes.RTLD_GLOBAL)
    return hip


def hip_runtimes_mapped():
    """Paths of every libamdhip64 mapped into this process (/proc/self/maps)."""
    try:
        with open("/proc/self/maps") as f:
            return {line.split()[-1] for line in f if "libamdhip64" in line}
    except OSError:  # pragma: no cover
        return set()


def load():
    """Load (once) and return the ctypes handle.  Raises PkError when the library is missing or when it would bring a
    second HIP runtime into the process."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise PkError(
                "libpk_amd.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`; "
                "this engine has no CPU fallback." % LIB_PATH)
        warn_unknown_switches()
        _map_torch_hip_runtime()
        lib = ctypes.CDLL(LIB_PATH)
        runtimes = hip_runtimes_mapped()
        if len(runtimes) > 1:
            raise PkError("two HIP runtimes in one process (%s): libpk_amd.so must share the one torch uses - streams and "
                          "device pointers of one runtime mean nothing to the other" % ", ".join(sorted(runtimes)))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().pk_last_error()
        raise PkError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


def raise_if_persist_failed():
    """Spin time-outs of the persistent recurrence land in a host-mapped counter."""
    lib = load()
    n = lib.pk_persist_error_count() + lib.pk_persist2_error_count()
    if n:
        lib.pk_persist_error_reset()
        lib.pk_persist2_error_reset()
        raise PkError("persistent recurrent kernel: %d wave(s) timed out waiting for a peer workgroup "
                      "(results of that launch are invalid)" % n)


# ----------------------------------------------------------------------------
# measurement: HIP-event timing of every C-ABI call (bench.py's roofline leg)
# ----------------------------------------------------------------------------
_profiler = None
_raw_load = load


class _TimedLib:
    """Proxy that brackets each entry point that takes a stream with two events recorded on
    torch's current stream - the stream the kernels are launched on."""

    def __init__(self, lib, prof):
        self._lib, self._prof = lib, prof

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        sig = SIGNATURES.get(name)
        if sig is None or not sig[1] or sig[1][0] is not P or name.startswith("pk_selftest_"):
            return fn
        prof = self._prof

        def timed(*a):
            import torch

            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            prof.events.append((name, e0, e1))
            return rc

        return timed


_EXP_CACHE = (None, {})
_KNOWN_SWITCHES = None
_warned_unknown = False


def _known_switches():
    """The PK_* names INTEGRATION.md documents (tests/test_env_switches_documented.py holds that table and the code to each
    other); read once, from the file that ships next to the package."""
    global _KNOWN_SWITCHES
    if _KNOWN_SWITCHES is None:
        import re

        try:
            doc = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
            _KNOWN_SWITCHES = frozenset(re.findall(r"`(PK_[A-Z0-9_]+)`", doc))
        except OSError:
            _KNOWN_SWITCHES = frozenset()
    return _KNOWN_SWITCHES


def warn_unknown_switches():
    """Once per process: PK_* variables in the environment that this version does not read.  The A/B levers of earlier
    rounds (PK_MLP_FUSED, PK_DIRECT_GRADS, PK_GEMM_SKINNY, ...) moved behind PK_EXPERIMENT=key=value; their old names would
    otherwise be ignored silently and an A/B script would compare two identical configurations."""
    global _warned_unknown
    if _warned_unknown:
        return
    _warned_unknown = True
    known = _known_switches()
    if not known:
        return
    # (PK_BENCH_* / PK_CPU_BASELINE / PK_REFERENCE / ... belong to bench.py, the tests and the golden generator)
    harness = ("PK_BENCH_", "PK_CPU_BASELINE", "PK_REFERENCE", "PK_FULL_SHAPE_JSON", "PK_GOLDEN_ONLY", "PK_DP_BACKEND")
    stray = sorted(k for k in os.environ if k.startswith("PK_") and k not in known and not k.startswith(harness))
    if stray:
        import warnings

        warnings.warn("pytorch-kaldi_amd: environment variable(s) %s are not switches of this version and are ignored (A/B "
                      "levers live behind PK_EXPERIMENT=key=value; INTEGRATION.md lists switches and keys)" % ", ".join(stray),
                      RuntimeWarning, stacklevel=3)


def experiment(key, default=None):
    """Value of one A/B lever of past experiments.  They live behind ONE environment variable,
    PK_EXPERIMENT="key=value,key=value" (the library reads its own keys from the same string: pk_lib.hip, same trimming of
    blanks; INTEGRATION.md lists them); a recipe author only ever needs the switches of INTEGRATION.md's first table.
    The string is parsed once per distinct value (this is called per layer and step)."""
    global _EXP_CACHE
    raw = os.environ.get("PK_EXPERIMENT", "")
    if _EXP_CACHE[0] != raw:
        d = {}
        for item in raw.split(","):
            k, sep, v = item.partition("=")
            if sep and k.strip() and k.strip() not in d:  # (the first occurrence wins, as in pk_lib.hip)
                d[k.strip()] = v.strip()
        _EXP_CACHE = (raw, d)
    return _EXP_CACHE[1].get(key, default)


class Profiler:
    def __init__(self):
        self.events = []

    def __enter__(self):
        global _profiler
        _profiler = self
        return self

    def __exit__(self, *exc):
        global _profiler
        _profiler = None

    def summary(self, steps):
        import torch

        torch.cuda.synchronize()
        acc = {}
        for name, e0, e1 in self.events:
            d = acc.setdefault(name, [0.0, 0])
            d[0] += e0.elapsed_time(e1)
            d[1] += 1
        return {k: {"ms_per_step": v[0] / steps, "calls_per_step": v[1] / steps, "avg_ms": v[0] / max(1, v[1])}
                for k, v in acc.items()}


class _TracedLib:
    """PK_DEBUG_CALLS=1 (fault hunting): every entry point that takes a stream prints its name before the call and the
    device is synchronised behind it - the last name on stderr is the launch a memory fault belongs to."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        sig = SIGNATURES.get(name)
        if sig is None or not sig[1] or sig[1][0] is not P:
            return fn

        def traced(*a):
            import sys

            import torch

            sys.stderr.write("[pk] %s\n" % name)
            sys.stderr.flush()
            rc = fn(*a)
            torch.cuda.synchronize()
            return rc

        return traced


def load():  # noqa: F811 - same name on purpose: every caller goes through the switch
    lib = _raw_load()
    if _profiler is not None:
        return _TimedLib(lib, _profiler)
    if os.environ.get("PK_DEBUG_CALLS", "0") == "1":
        return _TracedLib(lib)
    return lib
