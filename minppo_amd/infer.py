"""Runs inference for the trained model — same surface (and the same stub) as the reference's `minppo/infer.py`:
`load_model` unpickles the parameter tree that `save_model` wrote; `main` is not implemented upstream either
(`infer.py:22-27` raises NotImplementedError)."""

import logging
import pickle
import sys
from typing import Sequence

logger = logging.getLogger(__name__)


def load_model(filename: str) -> dict:
    with open(filename, "rb") as f:
        return pickle.load(f)


def main(args: Sequence[str] | None = None) -> None:
    """Runs inference with pretrained models."""
    if args is None:
        args = sys.argv[1:]
    raise NotImplementedError("Not implemented yet")


if __name__ == "__main__":
    main()
