"""A scene as a collection of submaps (GridNet) with per-submap SE(3) corrections
(reference: grid_opt/models/grid_atlas.py; visualisation methods are out of scope)."""
import logging
import math
from copy import deepcopy
from typing import Tuple

import contextlib

import torch
from torch import Tensor

from miso_amd import ops
import miso_amd.grid_opt.utils.utils as utils
import miso_amd.grid_opt.utils.utils_geometry as utils_geometry
from .base_net import BaseNet
from .grid_net import GridNet

logger = logging.getLogger(__name__)


class GridAtlas(BaseNet):
    def __init__(self, cfg: dict, device='cuda:0', dtype=torch.float32):
        super().__init__(cfg, device, dtype)
        self.cfg = cfg
        self.submaps = torch.nn.ModuleList()
        self.rotation_corrections = torch.nn.ParameterList()
        self.translation_corrections = torch.nn.ParameterList()
        self.R_world_submap_list = []
        self.t_world_submap_list = []
        self._submap_anchor_kf = []
        self._kf_id_to_submap_id = []
        self._submap_id_to_kf_ids = dict()
        self.curr_submap_id = -1
        self.curr_kf_id = -1

    # ---- locks ---------------------------------------------------------------------------
    def lock_submap(self, submap_id: int):
        s = self.get_submap(submap_id)
        s.lock_feature()
        s.lock_pose()

    def unlock_submap(self, submap_id: int):
        s = self.get_submap(submap_id)
        s.unlock_feature()
        s.unlock_pose()

    def _set_submap_pose_grad(self, flag: bool):
        for p in list(self.rotation_corrections) + list(self.translation_corrections):
            p.requires_grad_(flag)

    def lock_submap_pose(self):
        self._set_submap_pose_grad(False)

    def unlock_submap_pose(self):
        self._set_submap_pose_grad(True)

    def lock_keyframe_pose(self):
        for s in self.submaps:
            s.lock_pose()

    def unlock_keyframe_pose(self):
        for s in self.submaps:
            s.unlock_pose()

    def print_keyframe_pose_info(self):
        for i, s in enumerate(self.submaps):
            rot = math.degrees(torch.linalg.norm(s.rotation_corrections, dim=1).max())
            tran = torch.linalg.norm(s.translation_corrections.squeeze(2), dim=1).max()
            print(f"KF submap {i} pose corrections: max_rot={rot:.2f}deg, max_tran={tran:.2f}m")

    def print_submap_pose_info(self):
        for i in range(self.num_submaps):
            deg = math.degrees(torch.linalg.norm(self.rotation_corrections[i].detach()))
            print(f"Base submap {i} pose corrections: rot={deg:.2f}deg, "
                  f"tran={torch.linalg.norm(self.translation_corrections[i]):.2f}m")

    # ---- construction ----------------------------------------------------------------------
    def anchor_kf_for_submap(self, submap_id: int):
        return self._submap_anchor_kf[submap_id]

    def add_submap(self, local_bound: Tensor, Rws: Tensor, tws: Tensor, num_poses=1, optimize_poses=True):
        assert Rws.shape == (3, 3) and tws.shape == (3, 1)
        submap_id = len(self.submaps)
        cfg_model = deepcopy(self.cfg)
        cfg_model['grid']['bound'] = local_bound.numpy()
        cfg_model['pose']['num_poses'] = num_poses
        cfg_model['pose']['optimize'] = optimize_poses
        self.submaps.append(GridNet(cfg=cfg_model, device=self.device, dtype=self.dtype))
        self.R_world_submap_list.append(Rws.to(self.device))
        self.t_world_submap_list.append(tws.to(self.device))
        anchor_kf = self.curr_kf_id + 1          # the next keyframe anchors the new submap
        self._submap_anchor_kf.append(anchor_kf)
        self.rotation_corrections.append(torch.nn.Parameter(torch.zeros(1, 3, device=self.device)))
        self.translation_corrections.append(torch.nn.Parameter(torch.zeros(3, 1, device=self.device)))
        self.active_submaps = range(self.num_submaps)
        self.curr_submap_id = submap_id
        self._submap_id_to_kf_ids[submap_id] = {anchor_kf}

    def add_kf(self, Rsk: Tensor, tsk: Tensor):
        assert Rsk.shape == (3, 3) and tsk.shape == (3, 1)
        assert self.curr_submap_id >= 0, "No submap is created yet. Create a submap first."
        submap_id = self.curr_submap_id
        kf_global = self.curr_kf_id + 1
        kf_local = kf_global - self.anchor_kf_for_submap(submap_id)
        self._kf_id_to_submap_id.append(submap_id)
        self.get_submap(submap_id).set_initial_kf_pose(kf_local, Rsk, tsk, kf_key=f'KF{kf_global}')
        self._submap_id_to_kf_ids[submap_id].add(kf_global)
        self.curr_kf_id = kf_global
        return kf_global

    def set_kf_pose(self, kf_id: int, Rsk: Tensor, tsk: Tensor):
        assert Rsk.shape == (3, 3) and tsk.shape == (3, 1)
        submap_id = self.submap_id_for_kf(kf_id)
        kf_local = kf_id - self.anchor_kf_for_submap(submap_id)
        self.get_submap(submap_id).set_initial_kf_pose(kf_local, Rsk, tsk, kf_key=f'KF{kf_id}')

    def set_submap_pose(self, submap_id: int, Rws: Tensor, tws: Tensor):
        assert Rws.shape == (3, 3) and tws.shape == (3, 1)
        with torch.no_grad():
            self.R_world_submap_list[submap_id].copy_(Rws.to(self.device))
            self.t_world_submap_list[submap_id].copy_(tws.to(self.device))
            self.rotation_corrections[submap_id].zero_()     # new base pose: corrections restart at 0
            self.translation_corrections[submap_id].zero_()

    def set_submap_pose_correction(self, submap_id: int, R_delta: Tensor, t_delta: Tensor):
        assert R_delta.shape == (1, 3) and t_delta.shape == (3, 1)
        with torch.no_grad():
            self.rotation_corrections[submap_id].copy_(R_delta)
            self.translation_corrections[submap_id].copy_(t_delta)

    def set_active_submaps(self, active_submaps):
        self.active_submaps = active_submaps

    # ---- bookkeeping -----------------------------------------------------------------------
    @property
    def num_submaps(self):
        return len(self.submaps)

    @property
    def num_active_submaps(self):
        return len(self.active_submaps)

    @property
    def num_keyframes(self):
        return self.curr_kf_id + 1

    @property
    def num_levels(self):
        return self.get_submap(0).num_levels

    def num_keyframes_in_submap(self, submap_id: int) -> int:
        return len(self._submap_id_to_kf_ids[submap_id])

    def submap_id_for_kf(self, kf_id: int):
        return self._kf_id_to_submap_id[kf_id]

    def submap_id_for_kf_batch(self, kf_ids: Tensor) -> Tensor:
        table = torch.tensor(self._kf_id_to_submap_id, device=kf_ids.device)
        return table[kf_ids]

    def get_submap(self, submap_id: int) -> GridNet:
        assert 0 <= submap_id < self.num_submaps
        return self.submaps[submap_id]

    def _apply(self, fn, *args, **kwargs):
        """``.to()`` / ``.cuda()`` also move the submap base poses, which live in plain lists (upstream leaves
        them behind, so an atlas pickled from one device cannot be used on another)."""
        out = super()._apply(fn, *args, **kwargs)
        self.R_world_submap_list = [fn(R) for R in self.R_world_submap_list]
        self.t_world_submap_list = [fn(t) for t in self.t_world_submap_list]
        if self.R_world_submap_list:
            self.device = self.R_world_submap_list[0].device
        return out

    # ---- poses -------------------------------------------------------------------------------
    def initial_submap_pose(self, submap_id: int) -> Tuple[Tensor, Tensor]:
        return self.R_world_submap_list[submap_id], self.t_world_submap_list[submap_id]

    def updated_submap_pose(self, submap_id: int, device=None) -> Tuple[Tensor, Tensor]:
        cache = self.__dict__.get('_pose_cache')
        key = None
        if cache is not None:
            dr, dt = self.rotation_corrections[submap_id], self.translation_corrections[submap_id]
            key = (submap_id, str(device), dr._version, dt._version, torch.is_grad_enabled())
            hit = cache.get(submap_id)
            if hit is not None and hit[0] == key:
                return hit[1], hit[2]
        R0, t0 = self.initial_submap_pose(submap_id)
        R, t = utils_geometry.apply_pose_correction(R=R0, t=t0, R_delta=self.rotation_corrections[submap_id],
                                                    t_delta=self.translation_corrections[submap_id])
        if device is not None:
            R, t = R.to(device), t.to(device)
        if cache is not None:
            cache[submap_id] = (key, R, t)
        return R, t

    def updated_submap_poses_all(self, device=None) -> Tuple[Tensor, Tensor]:
        """All updated submap poses at once, (S,3,3) and (S,3,1): one batched exponential map (and one
        backward through it) instead of one per submap."""
        from miso_amd.so3 import so3_exp_map
        R0 = torch.stack(list(self.R_world_submap_list))
        t0 = torch.stack(list(self.t_world_submap_list))
        dr = torch.cat(list(self.rotation_corrections), dim=0)
        dt = torch.stack(list(self.translation_corrections))
        R, t = R0 @ so3_exp_map(dr), t0 + dt
        if device is not None:
            R, t = R.to(device), t.to(device)
        return R, t

    @contextlib.contextmanager
    def pose_cache(self):
        """Within the block, updated_submap_pose(s) is evaluated once per submap and parameter version
        and shared (graph included) by every pair that uses it.  For loops that sum the pair losses and
        call backward ONCE per iteration, like generic_align_multiple_submaps: the exponential map and
        its backward then run S times per iteration instead of 2 * pairs times."""
        prev = self.__dict__.get('_pose_cache')
        self.__dict__['_pose_cache'] = {}
        try:
            yield self
        finally:
            self.__dict__['_pose_cache'] = prev

    def _local_kf(self, kf_id: int, submap_id: int) -> int:
        expect = self.submap_id_for_kf(kf_id)
        assert expect == submap_id, f"Wrong submap for KF {kf_id}! Expect {expect}, got {submap_id}."
        return kf_id - self.anchor_kf_for_submap(submap_id)

    def initial_kf_pose_in_submap(self, kf_id: int, submap_id: int):
        return self.get_submap(submap_id).initial_kf_pose(self._local_kf(kf_id, submap_id))

    def updated_kf_pose_in_submap(self, kf_id: int, submap_id: int):
        return self.get_submap(submap_id).updated_kf_pose(self._local_kf(kf_id, submap_id))

    def initial_kf_pose_in_world(self, kf_id: int):
        s = self.submap_id_for_kf(kf_id)
        return utils_geometry.transform_poses_to(*self.initial_kf_pose_in_submap(kf_id, s),
                                                 *self.initial_submap_pose(s))

    def updated_kf_pose_in_world(self, kf_id: int):
        s = self.submap_id_for_kf(kf_id)
        return utils_geometry.transform_poses_to(*self.updated_kf_pose_in_submap(kf_id, s),
                                                 *self.updated_submap_pose(s))

    def global_bound(self, device='cpu') -> Tensor:
        corners = []
        for s in range(self.num_submaps):
            R, t = self.updated_submap_pose(s)
            b = self.get_submap(s).bound.to(device)
            xs, ys, zs = torch.meshgrid(b[0], b[1], b[2], indexing='ij')
            pts = torch.stack((xs, ys, zs), dim=-1).reshape(-1, 3)
            corners.append(utils_geometry.transform_points_to(pts, R.to(device), t.to(device)))
        corners = torch.cat(corners, dim=0)
        return torch.stack((corners.min(dim=0).values, corners.max(dim=0).values), dim=1)

    # ---- level / feature management ------------------------------------------------------------
    def ignore_level(self, l):
        for s in self.submaps:
            s.ignore_level(l)

    def include_level(self, l):
        for s in self.submaps:
            s.include_level(l)

    def zero_features(self):
        for s in self.submaps:
            s.zero_features()

    # ---- queries (hot path) ----------------------------------------------------------------------
    def _fused_query(self, x_world=None, axes=None, want_sdf=True, want_feats=False):
        """query_feature / forward as ONE launch (ops.AtlasQuery -> miso_atlas_sdf_fwd) when nothing has to be
        differentiated: the per-submap loop below encodes every point in every submap and moves (N,F) temporaries
        through HBM per submap; the kernel tests the bound first, encodes where a point is inside, keeps sum, count and
        mean in registers and decodes them on the spot.  None when the query is not eligible (autograd on, host
        tensors, a decoder or grid shape the fused kernels do not cover): the caller then runs the loop."""
        if torch.is_grad_enabled() or not self.active_submaps:
            return None
        probe = x_world if x_world is not None else self.get_submap(self.active_submaps[0]).features[0].feature
        if not probe.is_cuda:
            return None
        subs = [self.get_submap(s) for s in self.active_submaps]
        for sm in subs:
            for g in sm.features:
                # (a pickle the reference wrote holds NCDHW features; the kernels read channels-last rows: re-laid once)
                if g.feature.shape[1] > 1 and g.feature.stride(1) != 1:
                    g.feature.data = g.feature.data.contiguous(memory_format=torch.channels_last_3d)
        pack = None
        if want_sdf:
            pack = self.submaps[0]._fused_decoder()
            if pack is None:
                return None
        feats = [[g.feature for g in sm.features] for sm in subs]
        metas = [sm.features[0].grid_meta() for sm in subs]
        ekey = (id(pack), tuple(f.data_ptr() for fs in feats for f in fs))
        if self.__dict__.get('_atlas_eligible', (None, False))[0] != ekey:
            ok = all(ops.sdf_fused_supported(f, m, pack) if pack is not None else (f[0].shape[1] % 4 == 0)
                     for f, m in zip(feats, metas)) and len({(len(f), f[0].shape[1]) for f in feats}) == 1
            self.__dict__['_atlas_eligible'] = (ekey, ok)
        if not self.__dict__['_atlas_eligible'][1]:
            return None
        # the pose table, rebuilt when a correction (or an initial pose) changed: S exponential maps and ~10 small
        # launches per submap otherwise, per query
        act = tuple(self.active_submaps)
        pkey = (act,) + tuple((self.rotation_corrections[s]._version, self.translation_corrections[s]._version,
                               self.R_world_submap_list[s].data_ptr(), self.R_world_submap_list[s]._version,
                               self.t_world_submap_list[s].data_ptr(), self.t_world_submap_list[s]._version) for s in act)
        hit = self.__dict__.get('_atlas_poses')
        if hit is not None and hit[0] == pkey and hit[1].device == probe.device:
            poses = hit[1]
        else:
            rows = []
            for s in act:
                R, t = self.updated_submap_pose(s)
                # transfrom_points_from (utils_geometry.py:227-240): R_src_dst = R^T, t_src_dst = -R^T t
                Rinv = R.T
                rows.append(torch.cat((Rinv.reshape(-1), (-(Rinv @ t)).reshape(-1))))
            poses = torch.stack(rows).to(device=probe.device, dtype=torch.float32).contiguous()
            self.__dict__['_atlas_poses'] = (pkey, poses)
        q = self.__dict__.setdefault('_atlas_query', ops.AtlasQuery())
        try:
            return q(feats, metas, poses, pack, x=x_world, axes=axes, want_sdf=want_sdf, want_feats=want_feats)
        except RuntimeError as e:              # a shape outside the kernel table: the loop serves it
            if "not covered" in str(e):
                return None
            raise

    def sdf_on_lattice(self, xs: Tensor, ys: Tensor, zs: Tensor):
        """forward() on the meshgrid(xs, ys, zs, indexing='ij') lattice as an (nx, ny, nz) volume, the points generated
        inside the kernel (utils_sdf.extract_fields_device asks for this); None when the fused query is not eligible."""
        if int(xs.numel()) * int(ys.numel()) * int(zs.numel()) >= 2 ** 31:
            return None
        with torch.no_grad():
            got = self._fused_query(axes=(xs, ys, zs))
        return None if got is None else got[0].view(xs.numel(), ys.numel(), zs.numel())

    def query_feature(self, x_world: Tensor):
        """Mean over the active submaps that contain the point of that submap's features
        (reference :374-391; ignore_level is not applied here, as upstream)."""
        got = self._fused_query(x_world, want_sdf=False, want_feats=True)
        if got is not None:
            return got[1]
        total = 0
        count = 0
        for s in self.active_submaps:
            submap = self.get_submap(s)
            R, t = self.updated_submap_pose(s)
            x_local = utils_geometry.transfrom_points_from(x_world, R, t)
            inside = utils_geometry.coords_in_bound(x_local, submap.bound)
            total = total + inside * utils.grid_interp_regular(submap.features, x_local, ignore_level=None)
            count = count + inside
        count = count.float()
        count[count == 0] = 1
        return total / count

    def forward(self, x_world: Tensor, noise_std=0):
        got = self._fused_query(x_world)
        if got is not None:
            pred = got[0]
        else:
            feats = self.query_feature(x_world)
            pred = utils.grid_decode(feats, None, self.submaps[0].decoder, True)
        if noise_std > 0:
            pred = pred + torch.randn(pred.shape, device=x_world.device) * noise_std
        return pred

    def check_submap_intersection(self, src_id: int, dst_id: int, overlap_thresh=1e-2):
        """True if more than overlap_thresh of src's finest-level voxel centres fall inside
        dst's bound under the current poses (reference :405-420)."""
        src, dst = self.get_submap(src_id), self.get_submap(dst_id)
        pts = self._finest_vertices(src_id).to(self.device)
        R_s, t_s = self.updated_submap_pose(src_id)
        R_d, t_d = self.updated_submap_pose(dst_id)
        if pts.is_cuda:
            # one pass over the vertices, no (N,3) temporaries; the result stays on the device
            from miso_amd import ops
            bound = dst.features[0].grid_meta()
            cnt = ops.overlap_count(R_s, t_s, R_d, t_d, pts, [[bound.bound_min[a], bound.bound_max[a]] for a in range(3)])
            return (cnt / pts.shape[0]) > overlap_thresh
        world = utils_geometry.transform_points_to(pts, R_s, t_s)
        local = utils_geometry.transfrom_points_from(world, R_d, t_d)
        inside = utils_geometry.coords_in_bound(local, dst.bound)
        return (torch.count_nonzero(inside) / pts.shape[0]) > overlap_thresh

    def _finest_vertices(self, submap_id: int) -> Tensor:
        """Voxel centres of the finest level, built once per submap (the reference rebuilds
        the tensor on the host for every pair in every alignment iteration)."""
        cache = self.__dict__.setdefault('_vertex_cache', {})
        grid = self.get_submap(submap_id).features[-1]
        key = (submap_id, tuple(grid.feature.shape))
        if key not in cache:
            cache[key] = grid.vertex_positions().to(self.device)
        return cache[key]

    # ---- parameter groups ---------------------------------------------------------------------------
    def params_for_submap_pose(self, submap_id):
        return [self.rotation_corrections[submap_id], self.translation_corrections[submap_id]]

    def params_for_all_submap_poses(self):
        return [*self.rotation_corrections, *self.translation_corrections]

    def params_for_all_kf_poses(self):
        return [p for s in self.submaps for p in s.params_for_poses()]

    def params_for_all_features(self):
        return [p for s in self.submaps for p in s.params_for_features()]

    def params_at_level(self, level):
        return [p for s in self.submaps for p in s.params_at_level(level)]

    # ---- alignment coordinate cache -------------------------------------------------------------------
    def precompute_coordinates_for_alignment(self, norm_thresh=1e-5):
        """Per (submap, level): voxel centres whose multi-level feature norm exceeds the
        threshold, cached detached (reference :565-579)."""
        self._coords_for_alignment = dict()
        for level in range(self.num_levels):
            for s in range(self.num_submaps):
                submap = self.get_submap(s)
                coords = submap.features[level].vertex_positions().to(submap.device)
                with torch.no_grad():
                    norm = torch.linalg.norm(submap.query_feature(coords), dim=1)
                keep = norm > norm_thresh
                self._coords_for_alignment[f"submap{s}_level{level}"] = coords[keep].detach()
                # the same set in brick order for the fused pair stage (the reference's order stays what the API returns:
                # pairwise_loss_latent(subsample_points=...) draws vertex INDICES, align/miso.py:66-68)
                order = self._brick_order(submap.features[level].feature.shape[2:], keep)
                self._coords_for_alignment[f"submap{s}_level{level}_brick"] = coords[order].detach()

    @staticmethod
    def _brick_order(dims, keep: Tensor) -> Tensor:
        """Indices of the kept vertices of a (Z, Y, X) lattice (flat z-major index, as vertex_positions lists them) in BRICK
        order: 8 x 8 x 8 bricks in raster order, inside a brick its eight 4 x 4 x 4 sub-bricks, inside a sub-brick x fastest.
        The reference fixes the SET of alignment vertices, not their order (grid_atlas.py:565-579; every consumer sums over
        them).  The pair stage (csrc/pair_latent.hip) hands a wavefront 8 steps of 64 CONSECUTIVE vertices: in lattice order
        that is 2.5 lattice rows whose destination corner rows are wanted again by the rows and planes 200 and 20 000
        vertices later -- by another workgroup on another XCD, i.e. fetched again (counted: 2.1 x the compulsory bytes at
        cfg-4 level 1, FETCH_SIZE calibrated by tools/ubench/fetch_calib.hip); in brick order a wavefront's 512 vertices are
        one 8^3 block whose 9^3 destination corners it fetches once, and a step's 64 vertices one 4^3 cube with a tight box
        for the reach test (miso_align_src_boxes)."""
        import os
        Z, Y, X = (int(d) for d in dims)
        idx = torch.nonzero(keep.view(-1), as_tuple=False).view(-1)
        if os.environ.get("MISO_ALIGN_ORDER", "brick") != "brick" or idx.numel() == 0:
            return idx
        i, j, k = idx % X, (idx // X) % Y, idx // (X * Y)
        nbx, nby = (X + 7) // 8, (Y + 7) // 8
        brick = ((k >> 3) * nby + (j >> 3)) * nbx + (i >> 3)
        sub = (((k >> 2) & 1) * 2 + ((j >> 2) & 1)) * 2 + ((i >> 2) & 1)
        within = ((k & 3) * 4 + (j & 3)) * 4 + (i & 3)
        key = (brick * 8 + sub) * 64 + within
        return idx[torch.argsort(key)]

    def coordinates_for_alignment(self, submap_id: int, level: int, brick_order: bool = False):
        """brick_order: the same vertices ordered for the fused pair stage (_brick_order) -- for consumers that only sum over
        them; the default is the reference's lattice order."""
        assert 0 <= submap_id < self.num_submaps
        assert 0 <= level < self.num_levels
        key = f"submap{submap_id}_level{level}"
        if brick_order and (key + "_brick") in self.__dict__.get('_coords_for_alignment', {}):
            key += "_brick"
        if key not in self.__dict__.get('_coords_for_alignment', {}):
            raise ValueError(f"Coordinates for alignment not found for submap {submap_id} and level {level}. "
                             "Did you call precompute_coordinates_for_alignment()?")
        return self._coords_for_alignment[key]
