"""Free-space decomposition on the device, behind the reference's class interface.

Mirrors ``robotmpcs.utils.free_space_decomposition.FreeSpaceDecomposition``
(``free_space_decomposition.py:61-116``): ``set_position`` / ``compute_constraints`` /
``asdict`` / ``aslist`` for one seed, and ``compute_batch`` for the fleet case (B point clouds,
N seeds each -- the N calls per control step of ``examples/boxer_example.py:193-203`` and all
instances in one launch).  The work is done by ``rmpc_free_space_device`` (include/rmpc.h); there
is no CPU path.
"""
from __future__ import annotations

import numpy as np

from .._lib import free_space_decomposition_device


class FreeSpaceDecomposition:
    def __init__(self, number_constraints: int = 10, max_radius: float = 1.0, device: int = 0):
        self._number_constraints = int(number_constraints)
        self._max_radius = float(max_radius)
        self._device = int(device)
        self._position = np.zeros(3)
        self._planes = np.zeros((0, 4))

    def set_position(self, position):
        self._position = np.asarray(position, dtype=np.float64).reshape(3)

    def compute_batch(self, points, seeds):
        """points (B, P, 3) and seeds (B, N, 3), numpy or device tensors -> device tensor
        (B, N, number_constraints, 4) laid out as ``rmpc_scene.lin_constrs``."""
        import torch

        dev = torch.device("cuda", self._device)
        pts = torch.as_tensor(points, dtype=torch.float64).to(dev).contiguous()
        sds = torch.as_tensor(seeds, dtype=torch.float64).to(dev).contiguous()
        if pts.dim() != 3 or sds.dim() != 3 or pts.shape[0] != sds.shape[0] or pts.shape[2] != 3 or sds.shape[2] != 3:
            raise ValueError("points must be (B, P, 3) and seeds (B, N, 3)")
        out = torch.empty((pts.shape[0], sds.shape[1], self._number_constraints, 4), dtype=torch.float64, device=dev)
        free_space_decomposition_device(pts, sds, out, self._max_radius,
                                        stream=torch.cuda.current_stream(dev).cuda_stream)
        return out

    def compute_constraints(self, points):
        pts = np.asarray(points, dtype=np.float64).reshape(1, -1, 3)
        self._planes = self.compute_batch(pts, self._position.reshape(1, 1, 3))[0, 0].cpu().numpy()

    def constraints(self):
        return self._planes

    def asdict(self) -> dict:
        return {f"constraint_{i}": self._planes[i].copy() for i in range(self._planes.shape[0])}

    def aslist(self) -> np.ndarray:
        """(number_constraints, 4) array.  Unused slots hold the same dummy plane as ``asdict``; the
        reference's ``aslist`` (``:126-127``) builds its dummy with the two points swapped, which
        puts the plane through the seed itself (value 0 there) -- no example uses it and the
        intended far-away plane is returned instead."""
        return self._planes.copy()
