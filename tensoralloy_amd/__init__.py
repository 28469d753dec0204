"""
tensoralloy_amd — MI355X-native evaluator for TensorAlloy's hot path
(descriptor -> per-atom MLP -> energy -> analytic forces / virial), behind the
reference's own Python surface:

    from tensoralloy_amd import TensorAlloyCalculator          # tensoralloy.calculator
    from tensoralloy_amd.transformer import UniversalTransformer  # tensoralloy.transformer

All arithmetic runs in `libtensoralloy_amd.so` (hand-written HIP for gfx950,
C ABI in include/tensoralloy_amd.h). There is no CPU fallback.
"""
from .atoms import Atoms, HAVE_ASE
from .model import AtomicNN, SymmetryFunction, load_model
from .grap import GenericRadialAtomicPotential
from .transformer import UniversalTransformer, VirtualAtomMap
from .calculator import TensorAlloyCalculator
from .engine import Engine

__all__ = ["Atoms", "AtomicNN", "SymmetryFunction", "GenericRadialAtomicPotential", "UniversalTransformer", "VirtualAtomMap",
           "TensorAlloyCalculator", "Engine", "load_model", "HAVE_ASE"]
