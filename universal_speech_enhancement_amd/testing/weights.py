"""Deterministic, platform-independent random weights for the NCSN++ score network.

Why this exists
---------------
No trained checkpoint is reachable offline and a 65 M-parameter state dict (259 MB) cannot be
committed as a fixture, so every parity test and the benchmark regenerate the weights from a seed.
The generator is a counter-based integer hash (splitmix64) whose outputs are mapped to floats by an
exact dyadic division, so the same bits come out on every CPU / libm / numpy build: the GPU box
rebuilds exactly the tensors the golden vectors in ``tests/golden`` were produced with.

The recipe also defeats the reference's init degeneracy: with the reference's own initialisation
every ResBlock ``Conv_1``, the attention ``NIN_3`` and the pyramid convs are scaled by 1e-10
(``init_scale=0``; reference ncsnpp.py:59, layers.py:100-103, layerspp.py:74,273) so a random-init
network ignores its input.  Here every matrix-shaped tensor gets the fan-avg uniform distribution at
scale 1.0 (the reference's ``default_init(1.0)``, layers.py:66-103), biases are U(+-0.0866)
(std 0.05), GroupNorm affine parameters are perturbed away from (1, 0) so the affine path is
exercised, and the Fourier projection ``W`` is ~N(0, 16^2) (layerspp.py:35, ncsnpp.py:186).

Key names and shapes follow the reference ``NCSNpp.state_dict()`` layout
(``all_modules.<i>.<Conv_0|Conv_1|Conv_2|Dense_0|GroupNorm_0|GroupNorm_1|NIN_k>.<weight|bias|W|b>``,
``output_layer.weight|bias``; reference ncsnpp.py:116-316).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Sequence, Tuple

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, stream: str, n: int, lane: int = 0) -> np.ndarray:
    """n exact dyadic uniforms in [0, 1) (24-bit), float64, for (seed, stream-name, lane)."""
    key = np.uint64((zlib.crc32(stream.encode()) & 0xFFFFFFFF) | ((lane & 0xFFFF) << 32))
    base = _splitmix64(np.array([np.uint64(seed) ^ key], dtype=np.uint64))[0]
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + base)
    return (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def approx_normal(seed: int, stream: str, n: int) -> np.ndarray:
    """Irwin-Hall(12) - 6: unit-variance, exactly reproducible (sums of dyadic rationals)."""
    acc = np.zeros(n, dtype=np.float64)
    for lane in range(12):
        acc += uniform01(seed, stream, n, lane=lane + 1)
    return acc - 6.0


from ..sgmse.backbones.arch import ncsnpp_param_shapes  # noqa: E402,F401


def make_state_dict(seed: int = 1234, **arch) -> Dict[str, np.ndarray]:
    """Seeded float32 state dict (numpy) with reference key names for the given architecture."""
    shapes = ncsnpp_param_shapes(**arch)
    out: Dict[str, np.ndarray] = OrderedDict()
    for key, shape in shapes.items():
        n = int(np.prod(shape))
        leaf = key.rsplit(".", 1)[1]
        if key == "all_modules.0.W":
            v = approx_normal(seed, key, n) * 16.0
        elif len(shape) >= 2:
            if len(shape) == 4:
                rf = shape[2] * shape[3]
                fan_in, fan_out = shape[1] * rf, shape[0] * rf
            else:
                fan_in, fan_out = shape[1], shape[0]
            bound = np.sqrt(3.0 / ((fan_in + fan_out) / 2.0))
            v = (uniform01(seed, key, n) * 2.0 - 1.0) * bound
        elif leaf == "weight":  # a 1-D '.weight' is always a GroupNorm gamma
            v = 1.0 + 0.2 * (uniform01(seed, key, n) - 0.5)
        elif leaf == "bias" and _gn_bias(key, shapes):
            v = 0.2 * (uniform01(seed, key, n) - 0.5)
        else:  # conv / linear / NIN biases
            v = (uniform01(seed, key, n) * 2.0 - 1.0) * (0.05 * np.sqrt(3.0))
        out[key] = v.astype(np.float32).reshape(shape)
    return out


def _gn_bias(key: str, shapes) -> bool:
    """A 1-D '.bias' belongs to a GroupNorm iff its sibling '.weight' is 1-D too."""
    sib = key.rsplit(".", 1)[0] + ".weight"
    return sib in shapes and len(shapes[sib]) == 1


def weights_checksum(sd: Dict[str, np.ndarray]) -> str:
    """crc32 over all tensors' raw bytes in key order (hex); committed beside the golden vectors."""
    c = 0
    for k, v in sd.items():
        c = zlib.crc32(np.ascontiguousarray(v).tobytes(), c)
    return f"{c & 0xFFFFFFFF:08x}"


LARGE = dict(nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, input_channels=4)
LARGE_BOTH = dict(nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, input_channels=6)    # ScoreModel(condition="both")
SMALL12M = dict(nf=96, ch_mult=(1, 2, 2, 1), num_res_blocks=1, input_channels=4)      # NCSNpp12M (reference ncsnpp.py:527-541)
SMALL6M = dict(nf=96, ch_mult=(1, 1, 1, 1), num_res_blocks=1, input_channels=4)       # NCSNpp6M (reference ncsnpp.py:545-559)
# NCSNpp(discriminative=True): the generator of the LSGAN refine stage (reference ncsnpp.py:42-69 defaults + 86-92)
REFINE = dict(nf=128, ch_mult=(1, 2, 2, 2), num_res_blocks=1, input_channels=2, conditional=False)
