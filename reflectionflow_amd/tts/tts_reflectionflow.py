#!/usr/bin/env python3
"""MI355X counterpart of the reference's tts/tts_reflectionflow.py: entry point of cli.main("reflection")."""
from .cli import main as _main, parse_cli_args  # noqa: F401


def main(argv=None):
    return _main("reflection", argv)


if __name__ == "__main__":
    main()
