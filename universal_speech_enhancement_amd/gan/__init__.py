"""LSGAN refine stage: the generator that the reference runs after the SGMSE sampler."""
