"""`bean qc`: sample mask, replicate x guide outlier mask and guide editing rates - the step that
produces the hot path's inputs."""
from .sample_qc import fill_in_missing_samples, qc_masks  # noqa: F401
