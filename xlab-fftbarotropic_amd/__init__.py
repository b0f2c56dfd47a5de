"""MI355X-native hot path of XLab-FFTBarotropic: pseudospectral RK4 step behind a C ABI.

The directory name carries a hyphen (as the project name does); import it with
`importlib.import_module("xlab-fftbarotropic_amd")` or through the alias module
`xlab_fftbarotropic_amd` at the repository root.
"""
from .binding import FftBaroError, FftwfOperation, Model, lib, read_field, write_field, make_field, make_source_kuo2004, EXPORTS  # noqa: F401
from .build import build_lib  # noqa: F401
