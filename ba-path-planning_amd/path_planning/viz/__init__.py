from .plot_trajectories import plot_time_snapshots, plot_trajectories

__all__ = ["plot_trajectories", "plot_time_snapshots"]
