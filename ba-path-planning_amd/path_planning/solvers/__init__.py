"""Solvers of the MI355X-native planner: the drop-in ``SCP`` class (HIP kernels behind include/scp_hip.h)."""
from .scp import SCP  # noqa: F401

__all__ = ("SCP",)
