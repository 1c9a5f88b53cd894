"""crispr-bean_amd: MI355X-native implementation of the ``bean run`` SVI hot path.

Scope: the per-step ELBO + gradient + ClippedAdam loop that the
reference drives through Pyro in ``bean/model/run.py:347-396``, for the model
families in ``bean/model/model.py`` / ``bean/model/survival_model.py``.  The
compute lives in hand-written HIP (``csrc/``) behind a C ABI
(``include/bean_hip.h``); this package is the Python host side that mirrors the
reference's ``run_inference`` / ``identify_model_guide`` interface.

The directory name contains a hyphen, so import it either through the
``bean_amd`` shim at the repository root (``import bean_amd``) or with
``importlib.import_module("crispr-bean_amd")``.
"""

__version__ = "0.1.0"
