from .ReporterScreen import ReporterScreen, read_h5ad  # noqa: F401
from .Edit import (AminoAcidAllele, AminoAcidEdit, Allele, CodingNoncodingAllele, Edit, MutationType)  # noqa: F401
