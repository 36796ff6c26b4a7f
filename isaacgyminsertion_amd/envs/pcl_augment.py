"""Point-cloud augmentation applied by the environment before the PointNet sees the cloud
(isaacgyminsertion/tasks/factory_tactile/factory_utils.py:83-165: ``PointCloudAugmentations``): same method
names and argument meaning; every method is batched elementwise torch on the device the cloud lives on, so it
adds a handful of launches per environment step to the rollout, nothing to the update.  ``augment`` applies what
the reference applies (point-wise + per-env constant noise; rotation / outliers / dropout exist but are commented
out there, factory_utils.py:156-162)."""
import torch


class PointCloudAugmentations:
    def __init__(self, num_points=400, sigma=0.001, noise_clip=0.001, rotate_range=(-10, 10), scale_range=(0.8, 1.2),
                 dropout_ratio=0.2):
        self.num_points, self.sigma, self.noise_clip = num_points, sigma, noise_clip
        self.const_noise = 0.001
        self.rotate_range, self.scale_range, self.dropout_ratio = rotate_range, scale_range, dropout_ratio

    def random_noise(self, pointcloud_batch, pcl_noise, noise_prob=0.3):
        """clipped N(0, sigma) on a random 30 % of the points + a clipped per-env offset (in place, like the
        reference: factory_utils.py:94-101)."""
        B, N, _ = pointcloud_batch.shape
        jitter = (torch.randn_like(pointcloud_batch) * self.sigma).clamp_(-self.noise_clip, self.noise_clip)
        hit = (torch.rand(B, N, 1, device=pointcloud_batch.device) < noise_prob).to(pointcloud_batch.dtype)
        pointcloud_batch += jitter * hit
        return pointcloud_batch + (pcl_noise * self.const_noise).clamp(-self.noise_clip, self.noise_clip)

    def random_rotate(self, pointcloud_batch, angles_rad, axes):
        """row-vector points times the axis rotation of each env (axes: 0 = x, 1 = y, 2 = z)."""
        c, s = torch.cos(angles_rad), torch.sin(angles_rad)
        o, z = torch.ones_like(c), torch.zeros_like(c)
        rx = torch.stack([o, z, z, z, c, -s, z, s, c], -1)
        ry = torch.stack([c, z, s, z, o, z, -s, z, c], -1)
        rz = torch.stack([c, -s, z, s, c, z, z, z, o], -1)
        sel = axes.reshape(-1, 1)
        rot = torch.where(sel == 0, rx, torch.where(sel == 1, ry, torch.where(sel == 2, rz, torch.eye(
            3, device=c.device, dtype=c.dtype).reshape(1, 9).expand_as(rx))))
        return torch.bmm(pointcloud_batch, rot.reshape(-1, 3, 3))

    def random_scale_anisotropic(self, pointcloud_batch):
        lo, hi = self.scale_range
        f = torch.rand(pointcloud_batch.shape[0], 1, 3, device=pointcloud_batch.device) * (hi - lo) + lo
        return pointcloud_batch * f

    def add_outliers(self, pointcloud_batch, outlier_ratio=0.1, contour_prob=0.75, scale_factor=1.5):
        """replace a random 10 % of the points: 75 % just outside the cloud's bounding box, the rest N(0, 1.5)."""
        B, N, _ = pointcloud_batch.shape
        k, dev = int(N * outlier_ratio), pointcloud_batch.device
        lo = pointcloud_batch.min(dim=1, keepdim=True).values
        hi = pointcloud_batch.max(dim=1, keepdim=True).values
        beyond = torch.randn(B, k, 3, device=dev).abs()
        contour = torch.where(torch.rand(B, k, 3, device=dev) < 0.5, hi + beyond, lo - beyond)
        free = torch.randn(B, k, 3, device=dev) * scale_factor
        new = torch.where((torch.rand(B, k, 1, device=dev) < contour_prob), contour, free)
        idx = torch.randint(0, N, (B, k, 1), device=dev).expand(-1, -1, 3)
        return pointcloud_batch.scatter_(1, idx, new)

    def batch_random_dropout(self, coords, dropout_ratio=0.2, no_dropout_prob=0.8):
        """zero a random subset of points in the 20 % of clouds that are selected for dropout."""
        B, N, _ = coords.shape
        dev = coords.device
        if isinstance(dropout_ratio, float):
            ratios = torch.full((B, 1), dropout_ratio, device=dev)
        else:
            ratios = torch.rand(B, 1, device=dev) * (dropout_ratio[1] - dropout_ratio[0]) + dropout_ratio[0]
        exempt = torch.rand(B, 1, device=dev) >= 1 - no_dropout_prob
        drop = (torch.rand(B, N, device=dev) < ratios) & ~exempt
        return torch.where(drop.unsqueeze(-1), torch.zeros_like(coords), coords)

    def augment(self, pointcloud_batch, angle, axes, pcl_noise, dropout_ratio=0.2):
        if not pointcloud_batch.shape[0]:
            return pointcloud_batch
        return self.random_noise(pointcloud_batch, pcl_noise)
