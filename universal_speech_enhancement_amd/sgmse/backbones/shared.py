"""``BackboneRegistry`` -- same plug-in seam as the reference's ``sgmse/backbones/shared.py:10``."""
from ..util.registry import Registry

BackboneRegistry = Registry("Backbone")
