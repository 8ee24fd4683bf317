"""Inference input side: walks a folder for .wav files and yields the batch dict the reference's
``pad_to_longest_monaural_inference`` produces (``src/data/components/collate.py:42-73``): first channel, resampled to
``sampling_rate`` (FFT method), peak-normalised to 0.8 (``src/data/components/loadwav_dataset.py:90-120``), zero-padded to
the longest item.  Multi-process runs shard the file list over ranks like the reference's per-rank batches
(``src/data/loadwav_datamodule.py:53-60``)."""
from __future__ import annotations

import os
from typing import Dict, Iterator, List

import numpy as np
import torch

from .distributed import shard_list
from .wavio import load_utterance


class LoadWavData:
    def __init__(self, data_folder: str, target_folder: str, normalize: bool = True, sampling_rate: int = 24000,
                 batch_size: int = 1, num_workers: int = 0, rank: int = 0, world_size: int = 1, **ignored):
        self.data_folder, self.target_folder = data_folder, target_folder
        self.normalize, self.sampling_rate, self.batch_size = normalize, sampling_rate, batch_size
        files: List[str] = []
        for root, _, names in os.walk(data_folder):
            files += [os.path.join(root, n) for n in sorted(names) if n.endswith(".wav")]
        self.filepaths = shard_list(sorted(files), rank, world_size)

    def __len__(self):
        return len(self.filepaths)

    def _item(self, path: str) -> Dict:
        x, sr = load_utterance(path, self.sampling_rate, self.normalize)       # native loader (csrc/use_io.cpp)
        return {"perturbed": x, "name": os.path.basename(path).split(".wav")[0], "audio_path": path, "sampling_rate": sr}

    def predict_batches(self, device="cuda") -> Iterator[Dict]:
        for i in range(0, len(self.filepaths), self.batch_size):
            items = [self._item(p) for p in self.filepaths[i:i + self.batch_size]]
            lens = np.array([len(it["perturbed"]) for it in items], dtype=np.int32)
            wav = torch.zeros(len(items), int(lens.max()))
            for k, it in enumerate(items):
                wav[k, : lens[k]] = torch.from_numpy(it["perturbed"])
            yield {"perturbed": wav.to(device), "name": [it["name"] for it in items],
                   "sample_length": torch.from_numpy(lens), "sampling_rate": [it["sampling_rate"] for it in items],
                   "audio_path": [it["audio_path"] for it in items], "data_folder": self.data_folder,
                   "target_folder": self.target_folder}
