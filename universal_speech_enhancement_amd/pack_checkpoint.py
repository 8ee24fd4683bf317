"""Convert a Lightning checkpoint of the reference (``state_dict`` keys ``Score.score_net....`` for the SGMSE model,
``G.net....`` for the LSGAN refine generator) into the library's packed weight file:

    python -m universal_speech_enhancement_amd.pack_checkpoint ckpt=last.ckpt out=sgmse_large.usehip [model=SGMSE_Large|LSGAN] [precision=bf16|fp32]

The packing runs on the host (no GPU needed).  The file holds a versioned header (architecture, precision, blob layout
version, crc32) and the device blob as ``use_commit_weights`` would build it; ``predict ckpt_path=<file>.usehip`` and
``NCSNpp.load_weight_file`` start from it with a single read.  Re-pack after upgrading the library if the loader reports a
different blob layout version.
"""
from __future__ import annotations

import sys

import torch

from .hip_engine import HipScoreEngine
from .testing.weights import LARGE, REFINE

MODELS = {
    "SGMSE_Large": dict(arch=LARGE, prefixes=("Score.score_net.", "score_net.", ""), kw={}),
    "LSGAN": dict(arch=REFINE, prefixes=("G.net.", "net.", ""), kw=dict(input_channels=2, conditional=False, scale_by_sigma=False)),
}


def pack(state_dict: dict, out: str, model: str = "SGMSE_Large", precision: str = "bf16") -> str:
    m = MODELS[model]
    a = m["arch"]
    eng = HipScoreEngine(nf=a["nf"], ch_mult=a["ch_mult"], num_res_blocks=a["num_res_blocks"], precision=precision, device=0, **m["kw"])
    first = next(iter(eng.expected_weights()))
    prefix = next((p for p in m["prefixes"] if p + first in state_dict), None)
    if prefix is None:
        raise KeyError(f"no '{first}' under any of the prefixes {m['prefixes']} in the checkpoint")
    eng.set_weights(state_dict, prefix)
    eng.save_weight_blob(out)
    eng.close()
    return out


def main(argv=None):
    kv = dict(a.split("=", 1) for a in (sys.argv[1:] if argv is None else argv))
    if "ckpt" not in kv or "out" not in kv:
        raise SystemExit(__doc__)
    ckpt = torch.load(kv["ckpt"], map_location="cpu", weights_only=False)
    pack(ckpt.get("state_dict", ckpt), kv["out"], kv.get("model", "SGMSE_Large"), kv.get("precision", "bf16"))
    print("wrote", kv["out"])


if __name__ == "__main__":
    main()
