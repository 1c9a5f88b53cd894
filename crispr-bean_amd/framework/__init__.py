from .ReporterScreen import ReporterScreen, read_h5ad  # noqa: F401
