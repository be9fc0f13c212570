"""Command line interface: `python -m minppo_amd.cli {train,env,infer} <config> [dot.list=overrides]`
(same grammar as the reference's `minppo` console script, `minppo/cli.py:12-26`)."""

import argparse
import logging


def main() -> None:
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    parser = argparse.ArgumentParser(description="MinPPO CLI (MI355X engine)")
    parser.add_argument("command", choices=["train", "env", "infer"], help="Command to run")
    args, other_args = parser.parse_known_args()
    if args.command == "train":
        from minppo_amd.train import main as train_main

        train_main(other_args)
    elif args.command == "env":
        from minppo_amd.env import main as env_main

        env_main(other_args)
    elif args.command == "infer":
        from minppo_amd.infer import main as infer_main  # a stub upstream too (`minppo/infer.py:22-27` raises NotImplementedError)

        infer_main(other_args)
    else:
        raise ValueError(f"Invalid command: {args.command}")


if __name__ == "__main__":
    main()
