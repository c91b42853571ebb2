"""pytroy: the reference's Python package name (pybind/__init__.py does `from .pytroy_raw import *`)."""
from .pytroy_raw import *  # noqa: F401,F403
