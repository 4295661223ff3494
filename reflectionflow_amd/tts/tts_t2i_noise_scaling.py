#!/usr/bin/env python3
"""MI355X counterpart of the reference's tts/tts_t2i_noise_scaling.py: entry point of cli.main("noise_scaling")."""
from .cli import main as _main, parse_cli_args  # noqa: F401


def main(argv=None):
    return _main("noise_scaling", argv)


if __name__ == "__main__":
    main()
