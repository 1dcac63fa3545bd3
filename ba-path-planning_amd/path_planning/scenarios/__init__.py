"""Scenario generation: the reference layout (seed-reproducible) and the grid-swap family for large N."""
from .position_generator import generate_grid_swap, generate_positions, straight_line_min_distance  # noqa: F401

__all__ = ("generate_positions", "generate_grid_swap", "straight_line_min_distance")
