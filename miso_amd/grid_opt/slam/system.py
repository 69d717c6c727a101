"""Incremental SLAM driver: track the newest keyframe against the current submap, then map it
(reference: grid_opt/slam/system.py).

Orchestration only -- the time goes into ``Tracker.track`` (one SDF forward + coordinate backward +
``miso_lm_normal_eq`` per Gauss-Newton step) and ``Mapper.mapping`` (the captured trainer step).  The Open3D
visualiser and the marching-cubes mesh export of the reference are outside this path: pass any object with
``set_current_frame_points / update_geometries / update_view`` as ``visualizer`` to get the same call-backs.
"""
import logging
import os

import torch
from torch import Tensor

import miso_amd.grid_opt.utils.utils_geometry as utils_geometry
from miso_amd.grid_opt.datasets.submap_dataset import SubmapDataset
from miso_amd.grid_opt.models.grid_atlas import GridAtlas
from miso_amd.grid_opt.models.grid_net import GridNet
from miso_amd.grid_opt.slam.mapper import Mapper
from miso_amd.grid_opt.slam.tracker import Tracker

logger = logging.getLogger(__name__)


class System:
    def __init__(self, model: GridAtlas, dataset_track: SubmapDataset, dataset_map: SubmapDataset, cfg: dict,
                 R_world_origin: Tensor = None, t_world_origin: Tensor = None, verbose=True, visualizer=None):
        assert isinstance(model, GridAtlas), "Model must be an instance of GridAtlas."
        assert model.num_submaps == 0, "Input grid atlas is not empty."
        self.model, self.cfg, self.verbose = model, cfg, verbose
        self.dataset_track, self.dataset_map = dataset_track, dataset_map
        self.max_replay_frames = cfg['mapping']['max_replay_frames']
        self.max_replay_freq = cfg['mapping']['max_replay_freq']
        self.init_odom = cfg['system']['init_odom']
        self.log_dir = cfg['system']['log_dir']
        self.visualizer = visualizer
        # iterations of the first mapping of a submap / of every later keyframe (reference :97,:163 / :196)
        self.init_iterations, self.init_level_iterations = 50, 20
        self.kf_iterations, self.kf_level_iterations = 15, 5
        self.initialize_system(Rws=R_world_origin, tws=t_world_origin)

    def currrent_submap(self) -> GridNet:
        """The submap tracking and mapping work on (upstream spelling kept)."""
        return self.model.get_submap(self.model.curr_submap_id)

    current_submap = currrent_submap

    def current_kf_id(self) -> int:
        return self.model.curr_kf_id

    def _bind_submap(self):
        """A fresh tracker / mapper pair on the current submap, which is then initialised from its anchor keyframe."""
        # the submap left behind is not trained by this loop again: let go of the captured training plans kept with it
        # (gradient, Adam and binning buffers -- three quarters of a GB per ScanNet-sized submap)
        old = getattr(getattr(self, 'mapper', None), 'grid', None)
        if old is not None and old is not self.currrent_submap():
            old.__dict__.pop('_fast_plans', None)
        self.tracker = Tracker(model=self.currrent_submap(), dataset=self.dataset_track, cfg=self.cfg)
        self.mapper = Mapper(model=self.currrent_submap(), dataset=self.dataset_map, cfg=self.cfg)
        self.mapper.mapping(mapping_kfs=[self.current_kf_id()], iterations=self.init_iterations,
                            level_iterations=self.init_level_iterations)

    def _local_bound(self):
        return torch.tensor(self.cfg['system']['submap_local_bound'], dtype=torch.float32)

    def initialize_system(self, Rws: Tensor = None, tws: Tensor = None):
        """First submap at (Rws, tws), first keyframe at its origin, first mapping (reference :57-98)."""
        dev = self.cfg['device']
        Rws = (torch.eye(3) if Rws is None else Rws).to(dev)
        tws = (torch.zeros(3, 1) if tws is None else tws).to(dev)
        self.model.add_submap(self._local_bound(), Rws, tws, self.cfg['system']['submap_size'])
        self.model.add_kf(Rsk=torch.eye(3), tsk=torch.zeros(3, 1))
        self._bind_submap()

    def _odometry(self, src_id, like):
        if self.init_odom == 'external':
            return self.dataset_track.get_odometry_at_pose(src_id).to(like)
        if self.init_odom == 'static':
            return torch.eye(4).to(like)
        raise ValueError(f"Unknown odometry type: {self.init_odom}.")

    def initialize_next_kf_in_submap(self):
        """Next keyframe = previous keyframe composed with the odometry guess (reference :100-119)."""
        src_id = self.current_kf_id()
        with torch.no_grad():
            T_src = utils_geometry.pose_matrix(*self.model.updated_kf_pose_in_submap(src_id,
                                                                                      submap_id=self.model.curr_submap_id))
            T_dst = T_src @ self._odometry(src_id, T_src)
            self.model.add_kf(Rsk=T_dst[:3, :3], tsk=T_dst[:3, [3]])

    def should_create_new_submap(self) -> bool:
        s = self.cfg['system']
        if self.model.num_keyframes_in_submap(self.model.curr_submap_id) >= s['submap_size']:
            return True
        return self.tracker.latest_fov_overlap < s['submap_fov_thresh']

    def initialize_next_submap(self):
        """A new submap anchored at the odometry-propagated pose of the next keyframe, which sits at the new
        submap's origin (reference :128-165; the external odometry is used whatever ``init_odom`` says, as
        upstream)."""
        src_id = self.current_kf_id()
        with torch.no_grad():
            T_src = utils_geometry.pose_matrix(*self.model.updated_kf_pose_in_world(src_id))
            T_dst = T_src @ self.dataset_track.get_odometry_at_pose(src_id).to(T_src)
        self.model.add_submap(self._local_bound(), T_dst[:3, :3], T_dst[:3, [3]], self.cfg['system']['submap_size'])
        kf_id = self.model.add_kf(Rsk=torch.eye(3), tsk=torch.zeros(3, 1))
        assert kf_id == src_id + 1, "Keyframe ID mismatch."
        self._bind_submap()

    def replay_keyframes(self, first_frame_in_submap: int, head_kf: int):
        """Earlier keyframes of the submap mapped again with the newest one, against forgetting (reference :188-194)."""
        step = max((head_kf - first_frame_in_submap) // self.max_replay_frames, self.max_replay_freq)
        return list(range(first_frame_in_submap, head_kf, step)) + [head_kf]

    def step(self, first_frame_in_submap: int) -> int:
        """Consume one keyframe; returns the first keyframe of the (possibly new) current submap."""
        if self.should_create_new_submap():
            if self.cfg['system'].get('save_submap_mesh', False):
                from miso_amd.grid_opt.utils.utils_sdf import save_mesh
                submap = self.currrent_submap()
                save_mesh(submap, submap.bound, save_path=os.path.join(self.log_dir,
                                                                       f'submap_{self.model.curr_submap_id}.ply'),
                          resolution=256, device=self.cfg['device'])
            self.initialize_next_submap()          # the anchor keyframe of a new submap is not tracked
            return self.current_kf_id()
        self.initialize_next_kf_in_submap()
        head_kf = self.current_kf_id()
        self.tracker.track(optimize_kf=head_kf)
        self.mapper.mapping(mapping_kfs=self.replay_keyframes(first_frame_in_submap, head_kf),
                            iterations=self.kf_iterations, level_iterations=self.kf_level_iterations)
        if self.visualizer is not None:
            pts = self.dataset_track.sampled_points_at_kf(head_kf)
            self.visualizer.set_current_frame_points(pts.detach().cpu().numpy())
            self.visualizer.update_geometries(stop_frame=head_kf + 1)
            self.visualizer.update_view()
        return first_frame_in_submap

    def run(self):
        assert self.current_kf_id() == 0, "No keyframe exists. Did you call initialize_system()?"
        first = 0
        while self.model.num_keyframes != self.dataset_map.num_kfs:
            first = self.step(first)
