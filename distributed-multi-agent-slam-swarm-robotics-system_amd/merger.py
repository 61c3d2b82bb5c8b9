"""MapMerger: the ROS 2 node of server_nodes/map_merger.py without ROS.

`map_callback` follows map_merger.py:35-62 step by step on the GPU mapper's ops:
    grid_to_pcd (:64-85) -> first map adopted as global (:40-43) -> else ICP against the global cloud
    (threshold 1.0, identity init, point-to-point, 30 iterations, :45-52) -> fitness < 0.6 rejects
    (:54-56) -> transform, append, voxel_down_sample(resolution) (:58-60) -> publish_global_map (:87-127).
The registration / down-sampling arithmetic is Open3D's in the reference: PARITY UNPINNED here.
"""
import numpy as np


class MapMerger:
    def __init__(self, mapper, icp_threshold=1.0, icp_iterations=30, min_fitness=0.6):
        self.m = mapper
        self.global_xy = np.zeros((0, 2))          # self.global_pcd  :31
        self.map_resolution = 0.05                 # :32
        self.map_origin = [0.0, 0.0]               # :33
        self.icp_threshold, self.icp_iterations, self.min_fitness = icp_threshold, icp_iterations, min_fitness
        self.last_registration = None              # (T, fitness, rmse, iterations) of the last callback

    def map_callback(self, grid, resolution, origin_x, origin_y, agent_id=0):
        """One /agent_N/map message.  Returns (int8 global grid, (min_x, min_y)) or None when nothing is
        published (empty local map, or registration rejected)."""
        local = self.m.grid_to_pcd(grid, resolution, origin_x, origin_y)
        if len(local) == 0:                                                  # :37-38
            return None
        if len(self.global_xy) == 0:                                         # :40-43
            self.global_xy = local
            self.map_resolution = resolution
            self.map_origin = [origin_x, origin_y]
            self.last_registration = None
        else:
            T, fitness, rmse, it = self.m.icp(local, self.global_xy, self.icp_threshold, self.icp_iterations)
            self.last_registration = (T, fitness, rmse, it)
            if fitness < self.min_fitness:                                   # :54-56
                return None
            moved = local @ T[:2, :2].T + T[:2, 2]                           # local_pcd.transform  :58
            self.global_xy = self.m.voxel_downsample(np.concatenate([self.global_xy, moved]), self.map_resolution)  # :59-60
        return self.publish_global_map()

    def publish_global_map(self):                                            # :87-127
        if len(self.global_xy) == 0:
            return None
        return self.m.rasterise(self.global_xy, self.map_resolution)
