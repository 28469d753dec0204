"""Host-side mirror of `tensoralloy.transformer` (descriptor surface)."""
from .universal import UniversalTransformer
from .batch import BatchUniversalTransformer
from .vap import VirtualAtomMap
from .metadata import RadialMetadata, AngularMetadata

__all__ = ["UniversalTransformer", "BatchUniversalTransformer", "VirtualAtomMap", "RadialMetadata", "AngularMetadata"]
