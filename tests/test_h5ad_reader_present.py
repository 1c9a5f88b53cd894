"""Sentinel: the .h5ad ingestion tests (test_screen_io.py, test_qc.py, the GPU CLI tests) are skipped
when neither h5py nor the helper interpreter is present; this test says so loudly instead of letting
them vanish."""
import os

import bean_amd  # noqa: F401
from bean_amd.framework import h5ad_io


def test_an_h5ad_reader_is_available():
    try:
        import h5py  # noqa: F401
        return
    except ImportError:
        pass
    assert os.path.exists(h5ad_io.HELPER_PYTHON), (
        f"no h5py and no helper interpreter at {h5ad_io.HELPER_PYTHON} (set BEAN_H5PY_PYTHON): "
        "`bean run` cannot read .h5ad screens on this box and the ingestion tests are being skipped")
