"""What a deployment player needs from the robot side.  The reference talks to ``ExperimentEnv`` (ROS topics,
MoveIt, AllSight + RealSense drivers; algo/deploy/env/env.py) from inside ``HardwarePlayer``: observation assembly
(deploy_s1.py:364-471, deploy_s2.py:488-645) and actuation (deploy_s1.py:604-706) are robot code and out of scope
here (SURVEY.md section 8 f-4 names the checkpoint consumers only).  A player is handed an object with these four
methods; tensors live on the player's device, batch dimension 1 (or N for replayed logs)."""
import torch


class RobotIO:
    def observe(self):
        """-> dict with the entries the player's modalities need: 'obs' (B, numObservations), 'priv_info'
        (B, priv_info_dim) for the stage-1 policy; 'student_obs' (B, numObsStudent), 'tactile' (B, T, 3, H*W),
        'img' / 'seg' (B, T, H*W), 'pcl' (B, T, points*3) for the stage-2 student."""
        raise NotImplementedError

    def apply(self, action):
        """Send a clamped (B, 6) action: position / axis-angle displacement of the fingertip target."""
        raise NotImplementedError

    def done(self):
        """-> True when the episode is over (time-out or insertion, deploy_s1.py:473-491)."""
        raise NotImplementedError

    def reset(self):
        """Regrasp and move above the socket (deploy_s1.py:493-542)."""
        raise NotImplementedError


class ReplayRobot(RobotIO):
    """Feeds recorded observation dicts (a list, one per control tick) and keeps the actions it was sent: the
    stand-in used by the tests and for replaying logged trajectories through a restored policy."""

    def __init__(self, frames, episode_length=None):
        self.frames = list(frames)
        self.episode_length = episode_length or len(self.frames)
        self.t = 0
        self.actions = []

    def observe(self):
        return self.frames[self.t % len(self.frames)]

    def apply(self, action):
        self.actions.append(action.detach().clone())
        self.t += 1

    def done(self):
        return self.t > 0 and self.t % self.episode_length == 0

    def reset(self):
        pass

    def stacked_actions(self):
        return torch.stack(self.actions)
