from .position_generator import generate_grid_swap, generate_positions

__all__ = ["generate_positions", "generate_grid_swap"]
