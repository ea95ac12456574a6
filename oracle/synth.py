"""Seeded synthetic weights / inputs now live in the product package (``srbh_amd.synth``: bench.py and the harness need
them without touching the checker).  This shim keeps ``from oracle import synth`` working for the tests and tools."""
from srbh_amd.synth import *          # noqa: F401,F403
from srbh_amd.synth import _gen, _conv, _bn   # noqa: F401  (private helpers used by tools/make_golden.py)
