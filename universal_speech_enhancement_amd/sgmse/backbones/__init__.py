from .shared import BackboneRegistry
from .ncsnpp import NCSNpp, NCSNppLarge

__all__ = ["BackboneRegistry", "NCSNpp", "NCSNppLarge"]
