"""`minppo infer` - the reference's counterpart (`minppo/infer.py:17-27`) is `load_model` plus a `main` that raises
NotImplementedError; both are kept so that code written against the reference finds the same names.

`load_model` reads what `minppo_amd.train.save_model` (and the reference's `save_model`, `train.py:86-89`) writes: the pickled
nested parameter dict `{"params": {"MLP_0": {...}, "log_std": ..., "MLP_1": {...}}}`.  Unpickling executes whatever the file
contains - load only model files you wrote yourself (the engine's own checkpoints are pickle-free: `Trainer.load_checkpoint`)."""

from __future__ import annotations

import logging
import pickle
import sys
from typing import Sequence

logger = logging.getLogger(__name__)


def load_model(filename: str) -> dict:
    with open(filename, "rb") as f:
        return pickle.load(f)


def main(args: Sequence[str] | None = None) -> None:
    """Runs inference with pretrained models (not implemented upstream either: `minppo/infer.py:27`)."""
    if args is None:
        args = sys.argv[1:]
    raise NotImplementedError("Not implemented yet")


if __name__ == "__main__":
    main()
