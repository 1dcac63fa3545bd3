"""Host-side plots of a solved SCP instance (headless-safe; MPLBACKEND=Agg works).

Thin counterparts of SCP.visualize_trajectories / visualize_time_snapshots of the reference
(/root/reference/src/path_planning/solvers/scp.py:644-840): same call signatures on the solver, drawn from
the numpy copies in ``solver.trajectories``.  Plot styling is this repo's own; no numeric output."""
import numpy as np


def _axes_limits(solver):
    lo, hi = np.asarray(solver.pos_min, float), np.asarray(solver.pos_max, float)
    pos = solver.trajectories["positions"]
    lo = np.minimum(lo[:2], pos[..., :2].min(axis=(0, 1))) if np.all(np.isfinite(pos)) else lo[:2]
    hi = np.maximum(hi[:2], pos[..., :2].max(axis=(0, 1))) if np.all(np.isfinite(pos)) else hi[:2]
    return lo, hi


def plot_trajectories(solver, show=False, save_path="trajectories.pdf"):
    import matplotlib.pyplot as plt
    from matplotlib.patches import Circle

    pos = solver.trajectories["positions"]
    fig, ax = plt.subplots(figsize=(8, 8))
    cmap = plt.get_cmap("tab20")
    for i in range(solver.N):
        c = cmap(i % 20)
        ax.plot(pos[i, :, 0], pos[i, :, 1], "-", color=c, lw=1.2)
        ax.plot(pos[i, 0, 0], pos[i, 0, 1], "o", color=c, ms=5)
        ax.plot(pos[i, -1, 0], pos[i, -1, 1], "s", color=c, ms=5)
        ax.add_patch(Circle(pos[i, -1, :2], solver.R / 2, fill=False, color=c, lw=0.6))
    lo, hi = _axes_limits(solver)
    ax.set_xlim(lo[0], hi[0])
    ax.set_ylim(lo[1], hi[1])
    ax.set_aspect("equal")
    ax.set_xlabel("x [m]")
    ax.set_ylabel("y [m]")
    ax.set_title(f"SCP trajectories: N={solver.N}, K={solver.K}, R={solver.R}")
    if save_path:
        fig.savefig(save_path, bbox_inches="tight")
    if show:
        plt.show()
    return fig, ax


def plot_time_snapshots(solver, num_snapshots=5, save_path=None):
    import matplotlib.pyplot as plt
    from matplotlib.patches import Circle

    pos = solver.trajectories["positions"]
    ks = np.linspace(0, solver.K - 1, num_snapshots).astype(int)
    fig, axes = plt.subplots(1, num_snapshots, figsize=(4 * num_snapshots, 4), squeeze=False)
    cmap = plt.get_cmap("tab20")
    lo, hi = _axes_limits(solver)
    for ax, k in zip(axes[0], ks):
        for i in range(solver.N):
            c = cmap(i % 20)
            ax.plot(pos[i, : k + 1, 0], pos[i, : k + 1, 1], "-", color=c, lw=0.8, alpha=0.6)
            ax.add_patch(Circle(pos[i, k, :2], solver.R / 2, color=c, alpha=0.8))
        ax.set_xlim(lo[0], hi[0])
        ax.set_ylim(lo[1], hi[1])
        ax.set_aspect("equal")
        ax.set_title(f"t = {k * solver.h:.1f} s")
    if save_path:
        fig.savefig(save_path, bbox_inches="tight")
    return fig, axes
