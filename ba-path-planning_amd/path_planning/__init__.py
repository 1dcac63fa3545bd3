"""MI355X-native drop-in for jankammeth/BA-path-planning's ``path_planning`` package (hot path only)."""
from .scenarios.position_generator import generate_grid_swap, generate_positions

__all__ = ["SCP", "generate_positions", "generate_grid_swap"]


def __getattr__(name):  # SCP is imported lazily so that CPU-only tools can import the scenario generators
    if name == "SCP":
        from .solvers.scp import SCP

        return SCP
    raise AttributeError(name)
