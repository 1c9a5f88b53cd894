"""ctypes binding of ``libbean_hip.so`` (the C ABI in ``include/bean_hip.h``).

The library is built in-tree by ``build_library()`` (``hipcc
--offload-arch=gfx950``); ``load()`` fails loudly when it is missing - there is
no CPU fallback for the product path.
"""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_int32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.environ.get("BEAN_HIP_LIB") or os.path.join(LIB_DIR, "libbean_hip.so")  # env override: kernel A/B experiments
INCLUDE = os.path.join(os.path.dirname(_HERE), "include", "bean_hip.h")

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-shared",
    "-fPIC",
    "-munsafe-fp-atomics",
    # MachineLICM hoists the materialisation of 64-bit constants (two v_mov_b32 per float64 literal: a third of every
    # polynomial here) and other loop-invariant values out of the kernels' loops and keeps them in VGPRs across the whole
    # loop body: k_svi_async's item loop with its pieces inlined went to 448 B of scratch per lane (16 B without the
    # pass), and every guide kernel carries 6 - 22 more VGPRs with it.  Same arithmetic, same bits; measured on one box,
    # with / without the pass: metric 55.8 / 55.1 us per step on the two launches, 50.6 / 48.3 asynchronous, tiling
    # 153.5 / 151.9, tiling +Acc 171.5 / 168.4, survival 83.6 / 83.7.
    "-mllvm",
    "-disable-machine-licm",
]


class bean_hip_shape(ctypes.Structure):
    _fields_ = [
        ("family", c_int32),
        ("selection", c_int32),
        ("flags", c_int32),
        ("n_reps", c_int32),
        ("n_condits", c_int32),
        ("n_guides", c_int32),
        ("n_targets", c_int32),
        ("n_max_alleles", c_int32),
        ("n_edits", c_int32),
        ("n_ctrl", c_int32),
        ("mask_thres", c_int32),
        ("max_target_len", c_int32),
        ("guide_offset", c_int32),
        ("target_offset", c_int32),
        ("n_guides_total", c_int32),
        ("n_a2e_nnz", c_int32),
        ("sd_prior_scale", c_double),
        ("initial_lr", c_double),
        ("lrd", c_double),
        ("clip_norm", c_double),
        ("negctrl_loc", c_double),
        ("negctrl_scale", c_double),
        ("n_sample_covariates", c_int32),
        ("reserved_", c_int32),
        ("prior_ia_total", c_double),
    ]


# enum bean_hip_family / flags / buffer slots (include/bean_hip.h)
FAMILY = {"Normal": 0, "ControlNormal": 1, "MixtureNormal": 2, "MultiMixtureNormal": 3}
FLAG_USE_BCMATCH, FLAG_SCALE_BY_ACC, FLAG_FIT_NOISE, FLAG_PRIOR_NORMAL_MU, FLAG_DUMP_PI = 1, 2, 4, 8, 16
FLAG_NOT_LOSS_OWNER = 32
BUF = {
    "X": 0, "X_BC": 1, "ALLELE_CTRL": 2, "REPGUIDE": 3, "SIZE_FACTOR": 4, "SIZE_FACTOR_BC": 5,
    "SAMPLE_MASK": 6, "A0": 7, "A0_BC": 8, "PI_A0": 9, "Z_HI": 10, "Z_LO": 11,
    "TARGET_OFFSETS": 12, "GUIDE_TO_TARGET": 13, "ACCESSIBILITY": 14,
    "PRIOR_MU_LOC": 15, "PRIOR_MU_SCALE": 16, "PRIOR_SD_LOC": 17, "PRIOR_SD_SCALE": 18,
    "A2E_PTR": 19, "A2E_IDX": 20, "E2A_PTR": 21, "E2A_IDX": 22, "ALLELE_MASK": 23,
    "TIMEPOINTS": 24, "CONTROL_TIME": 25, "LOG_OBS0": 26, "NEGCTRL_MASK": 27,
    "XCHG_GSUM": 28, "XCHG_TGRAD": 29, "XCHG_SQ": 30, "REP_BY_COV": 31,
    "P": 32, "G": 48, "M": 64, "V": 80,
    "EPS_MU_IN": 96, "EPS_SD_IN": 97, "PI_IN": 98, "EPS_NOISE_IN": 99,
    "EPS_MU_OUT": 100, "EPS_SD_OUT": 101, "PI_OUT": 102, "EPS_NOISE_OUT": 103,
    "X0_IN": 104, "EPS_U_IN": 105, "X0_OUT": 106, "EPS_U_OUT": 107, "PRIOR_IA": 108, "XCHG_COV": 109, "GUIDE_IDS": 110,
    "LOSS_HIST": 112,
}
PARAM_ORDER = ("mu_loc", "mu_scale", "sd_loc", "sd_scale", "alpha_pi", "noise_loc", "noise_scale", "q0")

# every symbol include/bean_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("bean_hip_version", c_char_p, []),
    ("bean_hip_last_error", c_char_p, []),
    ("bean_hip_create", c_int32, [POINTER(bean_hip_shape), POINTER(c_void_p)]),
    ("bean_hip_destroy", c_int32, [c_void_p]),
    ("bean_hip_bind", c_int32, [c_void_p, c_int32, c_void_p, c_uint64]),
    ("bean_hip_prepare", c_int32, [c_void_p, c_void_p]),
    ("bean_hip_elbo_grad", c_int32, [c_void_p, c_uint64, c_uint64, c_uint64, c_void_p]),
    ("bean_hip_adam", c_int32, [c_void_p, c_uint64, c_void_p]),
    ("bean_hip_svi_run", c_int32, [c_void_p, c_uint64, c_uint64, c_uint64, c_int32, c_void_p]),
    ("bean_hip_svi_resume", c_int32, [c_void_p, c_uint64, c_uint64, c_uint64, c_int32, c_void_p]),
    ("bean_hip_sharded_begin", c_int32, [c_void_p, c_uint64, c_uint64, c_uint64, c_void_p]),
    ("bean_hip_sharded_sums", c_int32, [c_void_p, c_void_p]),
    ("bean_hip_sharded_guide", c_int32, [c_void_p, c_void_p]),
    ("bean_hip_sharded_update", c_int32, [c_void_p, c_int32, c_void_p]),
    ("bean_hip_comm_unique_id", c_int32, [c_char_p, c_void_p]),
    ("bean_hip_comm_init", c_int32, [c_void_p, c_char_p, c_void_p, c_int32, c_int32]),
    ("bean_hip_comm_all_reduce", c_int32, [c_void_p, c_void_p, c_uint64, c_void_p]),
    ("bean_hip_comm_destroy", c_int32, [c_void_p]),
    ("bean_hip_svi_run_exchanged", c_int32, [c_void_p, c_uint64, c_uint64, c_uint64, c_int32, c_void_p]),
    ("bean_hip_step_bytes", c_uint64, [c_void_p]),
    ("bean_hip_dominant_kernel", c_char_p, [c_void_p]),
    ("bean_hip_dominant_lds_bytes", c_uint64, [c_void_p]),
    ("bean_hip_dominant_kernel_variant", c_char_p, [c_void_p]),
    ("bean_hip_set_profile", c_int32, [c_void_p, c_int32]),
    ("bean_hip_get_profile", c_int32, [c_void_p, POINTER(c_double), POINTER(c_uint64)]),
    ("bean_hip_test_special", c_int32,
     [c_int32, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
]


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp"))] + [INCLUDE]


# Three builds of the same source: the kernels keep per-allele / per-condition state in registers and
# unrolled loops, so the number of alleles per guide and of conditions (sorting bins / timepoints)
# they hold are compile-time constants.  libbean_hip.so holds 8 of each (the fast path);
# libbean_hip_a16.so holds 16 of each and is loaded for the screens that need it; libbean_hip_a32.so holds 32
# alleles per guide (and 16 conditions) at two waves per SIMD, nothing spilled: 1.5 - 2.8 x faster than the
# allele-parallel kernels, which take over beyond 32 alleles (scripts/micro/tiling_sorted.py, TILING_AMAX).
AMAX_BUILDS = (8, 16, 32)
# A third build of the same source, libbean_hip_ab.so (-DBEAN_AB_KERNELS): the default library plus every
# superseded or opt-in kernel form (first wave form, split form, block forms, the one-launch step, the
# tile-persistent loop) that the BEAN_HIP_* switches select - the A/B references of DESIGN.md and of the
# bit-identity tests.  The product libraries contain the default kernels only.
AB = "ab"
ALL_BUILDS = (8, 16, 32, AB)


def tree_path(amax=8) -> str:
    """Where build_library() puts the build that holds ``amax`` alleles per guide: always in-tree."""
    if amax == 8:
        return os.path.join(LIB_DIR, "libbean_hip.so")
    if amax == AB:
        return os.path.join(LIB_DIR, "libbean_hip_ab.so")
    return os.path.join(LIB_DIR, f"libbean_hip_a{amax}.so")


def lib_path(amax=8) -> str:
    """What load() opens: the in-tree build unless BEAN_HIP_LIB / BEAN_HIP_LIB_A16 / _A32 name another build of
    that library (kernel A/B experiments).  The overrides are honoured by load() alone: building and the
    staleness check never touch a file outside the tree."""
    if amax == 8:
        return LIB_PATH
    if amax == AB:
        return tree_path(AB)
    return os.environ.get(f"BEAN_HIP_LIB_A{amax}") or tree_path(amax)


def is_stale(amax=8) -> bool:
    path = tree_path(amax)
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(s) > t for s in sources())


def build_library(force: bool = False, verbose: bool = False, amax=8) -> str:
    """Compile ``csrc/bean_hip.hip`` for gfx950 into ``lib/libbean_hip[_a16|_a32|_ab].so``."""
    path = tree_path(amax)
    if not force and not is_stale(amax):
        return path
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libbean_hip.so")
    os.makedirs(LIB_DIR, exist_ok=True)
    extra = [] if amax == 8 else (["-DBEAN_AB_KERNELS"] if amax == AB else
                                  [f"-DBEAN_AMAX={amax}", f"-DBEAN_BMAX={min(amax, 16)}"])
    # compiled beside the final path and moved there only after a clean scan: an unchecked build never sits
    # where load() (or the next staleness check) would take it for a good one
    tmp = path + ".tmp"
    cmd = [hipcc] + HIPCC_FLAGS + extra + [os.path.join(CSRC, "bean_hip.hip"), "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    try:
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
        # the toolchain's misplaced live-range copies (isa_check.py): a build that has one is not usable
        from . import isa_check

        bad = isa_check.findings_of(tmp)
        if bad:
            os.replace(tmp, path + ".rejected")
            lines = [f"{name} @ {addr:#x}: " + "; ".join(tx for _, tx in pre) for name, addr, _, pre in bad]
            raise RuntimeError(f"{path}: vector instructions in front of an exec restore at a control-flow join "
                               "(compiler fault, see crispr-bean_amd/isa_check.py):\n  " + "\n  ".join(lines))
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return path


_lib = None
_libs = {}


def load(amax=8):
    """Load the in-tree library (the build that holds ``amax`` alleles per guide); raises if it has
    not been built."""
    global _lib
    if amax == 8 and _lib is not None:
        return _lib
    if amax != 8:
        if amax not in _libs:
            import torch  # noqa: F401  (see below)

            path = lib_path(amax)
            if not os.path.exists(path):
                raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`. "
                                   "crispr-bean_amd has no CPU fallback.")
            lib = ctypes.CDLL(path)
            for name, res, args in SYMBOLS:
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _libs[amax] = lib
        return _libs[amax]
    # torch ships its own libamdhip64 (same SONAME as /opt/rocm's): import it
    # first so that this library binds to the one HIP runtime of the process.
    # Loaded the other way round, two runtimes coexist and device discovery
    # fails ("no ROCm-capable device is detected").
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "crispr-bean_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def rccl_path() -> str:
    """The RCCL shared object of this process: the copy PyTorch ships and has loaded (the library
    resolves ncclAllReduce ... from it at run time, ``bean_hip_comm_*``)."""
    import torch

    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return cand if os.path.exists(cand) else "librccl.so"


def check(status: int, what: str = "", lib=None):
    if status != 0:
        msg = (lib or load()).bean_hip_last_error().decode()
        raise RuntimeError(f"libbean_hip {what} failed: {msg}")
