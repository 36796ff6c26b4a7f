"""Learning side of the reference's deployment players (algo/deploy/deploy_s1.py, deploy_s2.py): checkpoint
consumers + the closed-loop policy step on the HIP kernels.  Robot I/O (ROS, MoveIt, cameras, AllSight drivers:
algo/deploy/env/*) is outside the hot path and enters through the small ``RobotIO`` protocol in ``robot_io.py``."""
from .robot_io import RobotIO, ReplayRobot  # noqa: F401
