"""smd_amd -- MI355X-native DDPM train + sample engine for symbolic-music latents.

Drop-in for the TransformerDDPM / DenseDDPM path of magenta/symbolic-music-diffusion
(train_ncsn.py / sample_ncsn.py): hand-written HIP kernels behind a C-ABI shared library
(include/smd_hip.h), Python host code mirroring the reference's callables.  Import of this package
never touches the GPU; the library is loaded on first use and there is no CPU fallback.
"""
__version__ = "0.1.0"

from . import flags, schedule  # noqa: F401  (CPU-only modules)

__all__ = ["flags", "schedule", "__version__"]
