"""Helpers the SED path needs (mirror of `psa.utils`, reference src/psa/utils/__init__.py)."""
from .helpers import parse_direction

__all__ = ["parse_direction"]
