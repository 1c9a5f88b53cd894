"""C-ABI library: loads, exports every symbol include/bean_hip.h declares, and
rejects bad arguments before touching a device (no GPU needed)."""
import ctypes
import os
import re

import pytest

import bean_amd  # noqa: F401
from bean_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build_library()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "bean_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bean_hip_[a-z_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in bean_hip.h but not exported"
    assert set(names) == {s[0] for s in _lib.SYMBOLS}


def test_version_string(lib):
    assert b"gfx950" in lib.bean_hip_version()


@pytest.mark.parametrize("amax", [a for a in _lib.ALL_BUILDS if a != 8])
def test_every_build_exports_the_header(amax):
    """The 16- and 32-allele builds and the A/B library are the same source: same symbols, their own version
    string (what `__graft_entry__.build()` leaves in crispr-bean_amd/lib/)."""
    path = _lib.lib_path(amax)
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} not built yet (python -c 'import __graft_entry__ as g; g.build()')")
    lib = _lib.load(amax)
    for n in declared_symbols():
        assert hasattr(lib, n), (amax, n)
    v = lib.bean_hip_version()
    assert b"gfx950" in v and (amax == _lib.AB or f"{amax} alleles".encode() in v)
    assert _lib.BUF["GUIDE_IDS"] == 110


def test_slot_numbers_match_header():
    text = open(os.path.join(ROOT, "include", "bean_hip.h")).read()
    body = text[text.index("enum bean_hip_buf"):]
    body = re.sub(r"/\*.*?\*/", "", body[: body.index("};")], flags=re.S)
    val, table = -1, {}
    for tok in body.split("{", 1)[1].split(","):
        tok = tok.strip()
        if not tok:
            continue
        if "=" in tok:
            name, v = [t.strip() for t in tok.split("=")]
            val = int(v)
        else:
            name, val = tok, val + 1
        table[name] = val
    assert table["BEAN_BUF_X"] == _lib.BUF["X"]
    assert table["BEAN_BUF_PRIOR_SD_SCALE"] == _lib.BUF["PRIOR_SD_SCALE"]
    assert table["BEAN_BUF_P_MU_LOC"] == _lib.BUF["P"]
    assert table["BEAN_BUF_G_MU_LOC"] == _lib.BUF["G"]
    assert table["BEAN_BUF_M_MU_LOC"] == _lib.BUF["M"]
    assert table["BEAN_BUF_V_MU_LOC"] == _lib.BUF["V"]
    assert table["BEAN_BUF_EPS_NOISE_OUT"] == _lib.BUF["EPS_NOISE_OUT"]
    assert table["BEAN_BUF_LOSS_HIST"] == _lib.BUF["LOSS_HIST"]
    assert table["BEAN_BUF_P_Q0"] - table["BEAN_BUF_P_MU_LOC"] == len(_lib.PARAM_ORDER) - 1
    assert table["BEAN_BUF_V_Q0"] - table["BEAN_BUF_V_MU_LOC"] == len(_lib.PARAM_ORDER) - 1
    for name in ("A2E_PTR", "ALLELE_MASK", "TIMEPOINTS", "LOG_OBS0", "NEGCTRL_MASK", "XCHG_GSUM", "XCHG_TGRAD",
                 "XCHG_SQ", "X0_IN", "EPS_U_OUT"):
        assert table["BEAN_BUF_" + name] == _lib.BUF[name], name


def _shape(**kw):
    base = dict(family=2, selection=0, flags=1, n_reps=2, n_condits=5, n_guides=10, n_targets=2,
                n_max_alleles=2, n_edits=0, n_ctrl=1, mask_thres=10, max_target_len=5, guide_offset=0,
                target_offset=0, n_guides_total=0, reserved=0, sd_prior_scale=0.01,
                initial_lr=0.01, lrd=0.999, clip_norm=10.0, negctrl_loc=0.0, negctrl_scale=0.1)
    base.update(kw)
    return _lib.bean_hip_shape(**base)


@pytest.mark.parametrize("kw,msg", [
    (dict(family=7), "family"),
    (dict(selection=2), "selection"),
    (dict(family=3, n_max_alleles=300, n_edits=2, n_targets=2), "n_max_alleles"),
    (dict(n_sample_covariates=2), "sample covariates"),
    (dict(family=0, n_sample_covariates=100), "n_sample_covariates"),
    (dict(family=3, n_max_alleles=4, n_edits=3, n_targets=2), "n_targets == n_edits"),
    (dict(n_condits=9), "n_condits"),
    (dict(n_guides=0), ">= 1"),
    (dict(n_max_alleles=3), "n_max_alleles == 2"),
    (dict(family=1, n_targets=2), "ControlNormal"),
    (dict(lrd=0.0), "lrd"),
    (dict(guide_offset=5, n_guides_total=12), "shard offsets"),
])
def test_create_rejects_bad_shapes_without_a_device(lib, kw, msg):
    h = ctypes.c_void_p()
    s = _shape(**kw)
    assert lib.bean_hip_create(ctypes.byref(s), ctypes.byref(h)) != 0
    assert msg in lib.bean_hip_last_error().decode()


def test_null_handles_are_errors(lib):
    assert lib.bean_hip_bind(None, 0, None, 0) != 0
    assert lib.bean_hip_prepare(None, None) != 0
    assert lib.bean_hip_svi_run(None, 0, 0, 1, 0, None) != 0
    assert lib.bean_hip_destroy(None) == 0


def test_missing_library_is_a_loud_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()
