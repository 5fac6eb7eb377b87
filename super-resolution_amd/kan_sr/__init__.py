"""kan_sr — MI355X-native ConvSepKAN hot path (HIP kernels behind a C ABI, PyTorch as plumbing)."""
from . import _lib, ops  # noqa: F401
