"""Logging with the reference's event points and levels.

The reference logs through ``log`` + ``env_logger`` controlled by ``RUST_LOG``
(/root/reference/src/main.rs:3, :352): protocol steps at debug (main.rs:93, 215, 224,
246, 249, 263, 281, 298, 303, 309), malformed verdicts at error (main.rs:150) and the
final answer at info (main.rs:464). The same events are emitted here on the
``consensus`` logger; the level comes from ``$CONSENSUS_LOG`` (``RUST_LOG`` is honoured
as a fallback so existing habits keep working). Unlike the reference, the final
answer is also printed to stdout by the CLI.
"""
from __future__ import annotations

import logging
import os
import sys

_LEVELS = {
    "trace": logging.DEBUG,
    "debug": logging.DEBUG,
    "info": logging.INFO,
    "warn": logging.WARNING,
    "warning": logging.WARNING,
    "error": logging.ERROR,
    "off": logging.CRITICAL + 10,
}

_configured = False


def get_logger(name: str = "consensus") -> logging.Logger:
    global _configured
    if not _configured:
        configure()
    return logging.getLogger(name)


def configure(level: str | None = None) -> None:
    global _configured
    spec = level or os.environ.get("CONSENSUS_LOG") or os.environ.get("RUST_LOG") or "error"
    spec = spec.split(",")[0].split("=")[-1].strip().lower()  # accept "crate=debug" forms
    root = logging.getLogger("consensus")
    root.setLevel(_LEVELS.get(spec, logging.ERROR))
    if not root.handlers:
        h = logging.StreamHandler(sys.stderr)
        h.setFormatter(logging.Formatter("[%(asctime)s %(levelname)s %(name)s] %(message)s", "%H:%M:%S"))
        root.addHandler(h)
    root.propagate = False
    _configured = True
