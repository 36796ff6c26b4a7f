"""MI355X-native learning-side hot path of osheraz/IsaacGymInsertion (teacher PPO update,
student encoders) behind the reference's own Python API.  See DESIGN.md / INTEGRATION.md."""
__version__ = "0.1.0"
