"""``compute-trajectories``: one demo solve + plots (entry point of the reference,
/root/reference/src/path_planning/cli/compute_trajectories.py:9-99, pyproject.toml:53).

Called without arguments it reproduces the reference demo (N=10, T=100 s, h=0.2 s -> K=500, R=0.8 m, space
200 x 200 m, generator scenario, max_iterations=15, two plots).  The optional flags expose what the
reference hard-codes (its SURVEY "next" item f-1): problem size, scenario family and seed, plot files."""
import argparse
import time

import numpy as np

from ..scenarios.position_generator import generate_grid_swap, generate_positions
from ..solvers.scp import SCP


def build_parser():
    p = argparse.ArgumentParser(prog="compute-trajectories", description=__doc__.split("\n\n")[0])
    p.add_argument("--n-agents", type=int, default=10)
    p.add_argument("--time-horizon", type=float, default=100.0)
    p.add_argument("--time-step", type=float, default=0.2)
    p.add_argument("--min-distance", type=float, default=0.8)
    p.add_argument("--space", type=float, nargs="+", default=None, help="[min..., max...]; default 0 0 200 200")
    p.add_argument("--scenario", choices=["reference", "grid-swap"], default="reference")
    p.add_argument("--dim", type=int, choices=[2, 3], default=2)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--max-iterations", type=int, default=15)
    p.add_argument("--polish", action="store_true",
                   help="one more joint QP at 1e-8 after the SCP loop: the result meets every constraint to ~1e-6")
    p.add_argument("--cg-iters", type=int, default=None,
                   help="PCG steps per ADMM step of the joint QP (scp_qp_settings.cg_iters; default 1)")
    p.add_argument("--no-plots", action="store_true")
    p.add_argument("--save-prefix", default=None, help="write <prefix>_2d.pdf and <prefix>_snapshots.pdf")
    return p


def main(argv=None):
    """Create collision-free trajectories for a set of initial and final positions."""
    args = build_parser().parse_args([] if argv is None else argv)
    print("------ WOW Fleet Collision-Free 2D Trajectory Generation ------")

    n_vehicles = args.n_agents
    time_horizon = args.time_horizon
    time_step = args.time_step
    min_distance = args.min_distance
    if args.scenario == "grid-swap":
        initial_positions, final_positions, space_dims = generate_grid_swap(
            n_vehicles, seed=args.seed or 0, dim=args.dim)
        if args.space is not None:
            space_dims = args.space
    else:
        space_dims = args.space if args.space is not None else [0, 0, 200, 200]
        initial_positions, final_positions = generate_positions(n_vehicles, min_distance, seed=args.seed)

    print("Configuration:")
    print(f"  Number of vehicles: {n_vehicles}")
    print(f"  Time horizon: {time_horizon} s")
    print(f"  Time step: {time_step} s")
    print(f"  Minimum margin: {min_distance} m")
    print(f"  Space dimensions: {space_dims} m")
    print()

    try:
        solver = SCP(
            n_vehicles=n_vehicles,
            time_horizon=time_horizon,
            time_step=time_step,
            min_distance=min_distance,
            space_dims=space_dims,
            dim=args.dim,
            polish=args.polish,
            qp_settings={"cg_iters": args.cg_iters} if args.cg_iters else None,
        )
        print(f"Successfully generated positions for {n_vehicles} vehicles")
        solver.set_initial_states(np.asarray(initial_positions))
        solver.set_final_states(np.asarray(final_positions))

        print("Generating trajectories...")
        start_time = time.time()
        solver.generate_trajectories(max_iterations=args.max_iterations)
        end_time = time.time()

        print("\nTrajectory generation complete!")
        print(f"Total computation time: {end_time - start_time:.3f} seconds")
        print(f"Number of time steps: {solver.K}")
        print(f"Total trajectory duration: {solver.T} seconds")

        if not args.no_plots:
            pre = args.save_prefix
            print("\nVisualizing 2D trajectories...")
            solver.visualize_trajectories(show_animation=pre is None,
                                          save_path=f"{pre}_2d.pdf" if pre else "trajectories.pdf")
            print("\nVisualizing time snapshots")
            solver.visualize_time_snapshots(num_snapshots=5, save_path=f"{pre}_snapshots.pdf" if pre else None)
        return solver
    except Exception as e:  # the reference swallows and prints every error (compute_trajectories.py:98-99)
        print(f"Error during trajectory generation: {e}")
        return None


def console_main():
    import sys

    main(sys.argv[1:])


if __name__ == "__main__":
    console_main()
