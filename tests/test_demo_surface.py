"""The import surface of the reference's demos (north-star: demo/build_submaps.py and demo/align_submaps.py run
unmodified).  With ``miso_amd.compat`` installed -- ``grid_opt`` and ``cuda_gridsample`` resolve to this package -- and
test-only stand-ins for the two GUI / evaluation packages the demos import at module top (open3d, evo: not in the
image and outside the hot path), each demo's import block executes and every global name its functions use resolves.
The demo sources are read from the reference checkout, which exists in the build container only: skipped elsewhere.
tools/demo_synthetic.py then runs the two demos' call sequence end to end on synthetic frames (GPU test below)."""
import ast
import builtins
import os
import subprocess
import sys
import types

import pytest

REF = os.environ.get("MISO_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub(name):
    class _Any(types.ModuleType):
        def __getattr__(self, attr):
            if attr.startswith("__"):
                raise AttributeError(attr)
            sub = _Any(f"{self.__name__}.{attr}")
            setattr(self, attr, sub)
            return sub

        def __call__(self, *a, **k):
            return self
    m = _Any(name)
    m.__path__ = []
    return m


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "demo")), reason="reference checkout not present")
@pytest.mark.parametrize("demo", ["build_submaps.py", "align_submaps.py"])
def test_demo_import_block_and_globals_resolve(demo, monkeypatch):
    import miso_amd.compat  # noqa: F401
    for name in ("open3d", "evo", "evo.core", "evo.core.metrics"):
        monkeypatch.setitem(sys.modules, name, _stub(name))
    sys.modules["evo"].core = sys.modules["evo.core"]
    sys.modules["evo.core"].metrics = sys.modules["evo.core.metrics"]
    src = open(os.path.join(REF, "demo", demo)).read()
    tree = ast.parse(src)
    imports = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    ns = {"__name__": "demo_under_test"}
    exec(compile(ast.Module(body=imports, type_ignores=[]), demo, "exec"), ns)
    # module-level definitions of the script itself
    defined = set(ns) | {n.name for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef))}
    for node in tree.body:
        if isinstance(node, ast.Assign):
            defined |= {t.id for t in node.targets if isinstance(t, ast.Name)}
    missing = {}
    for fn in (n for n in tree.body if isinstance(n, ast.FunctionDef)):
        local = {a.arg for a in fn.args.args + fn.args.kwonlyargs}
        for node in ast.walk(fn):
            if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Store):
                local.add(node.id)
            elif isinstance(node, (ast.For, ast.comprehension)):
                local |= {t.id for t in ast.walk(node.target) if isinstance(t, ast.Name)}
            elif isinstance(node, ast.withitem) and node.optional_vars is not None:
                local |= {t.id for t in ast.walk(node.optional_vars) if isinstance(t, ast.Name)}
        for node in ast.walk(fn):
            if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Load):
                if node.id not in local and node.id not in defined and not hasattr(builtins, node.id):
                    missing.setdefault(fn.name, set()).add(node.id)
    assert not missing, f"{demo}: names the mirror does not provide: {missing}"
    # the attributes the demos reach for on the imported modules
    for mod, attrs in (("utils_scannet", ("scannet_scenes", "create_scannet_dataset")),
                       ("utils_sdf", ("save_mesh",)), ("utils_eval", ("evo_trajectory_error",)),
                       ("utils_vis", ("beautiful_rgb",))):
        if mod in ns:
            for a in attrs:
                assert hasattr(ns[mod], a), (mod, a)
    for a in ("identity_rotations", "chordal_to_degree", "wrapped_gaussian_rotations", "gaussian_translations",
              "pose_matrix", "transform_points_to", "transfrom_points_from"):
        assert hasattr(ns["utils_geometry"], a), a
    assert ns["utils_scannet"].scannet_scenes()["0000_00"].num_kfs == 372


def test_trajectory_error_metrics():
    """utils_eval.evo_trajectory_error: zero after alignment for a rigidly moved copy; translation / rotation parts of a
    known perturbation (evo's definitions: |t| of inv(P) Q, Frobenius norm of R_err - I)."""
    import numpy as np
    import torch
    import miso_amd.grid_opt.utils.utils_eval as E
    import miso_amd.grid_opt.utils.utils_geometry as G
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_cases as gc
    rs = np.random.RandomState(0)
    n = 6
    R1 = torch.tensor(np.stack([gc.rodrigues(rs.uniform(-1, 1, 3)) for _ in range(n)]), dtype=torch.float32)
    t1 = torch.tensor(rs.uniform(-3, 3, (n, 3, 1)), dtype=torch.float32)
    Rg = torch.tensor(gc.rodrigues([0.3, -0.5, 0.2]), dtype=torch.float32)
    tg = torch.tensor([[1.0], [-2.0], [0.5]])
    R2, t2 = Rg @ R1, Rg @ t1 + tg                                   # the same trajectory in another world frame
    for rel in (E.PoseRelation.translation_part, E.PoseRelation.rotation_part):
        st = E.evo_trajectory_error(R1, t1, R2, t2, pose_relation=rel, align=True).get_all_statistics()
        assert st["rmse"] < 1e-5 and set(st) == {"rmse", "mean", "median", "std", "min", "max", "sse"}
    assert E.evo_trajectory_error(R1, t1, R2, t2, align=False).get_all_statistics()["rmse"] > 0.5
    # a pure rotation error of 10 degrees on every pose: chordal distance 2 sqrt(2) sin(theta / 2)
    Rd = torch.tensor(gc.rodrigues([0.0, 0.0, np.radians(10.0)]), dtype=torch.float32)
    st = E.evo_trajectory_error(R1, t1, R1 @ Rd, t1, pose_relation=E.PoseRelation.rotation_part, align=False)
    chord = st.get_all_statistics()["rmse"]
    assert abs(chord - 2 * np.sqrt(2) * np.sin(np.radians(5.0))) < 1e-5
    assert abs(float(G.chordal_to_degree(chord)) - 10.0) < 1e-2


@pytest.mark.gpu
def test_synthetic_demo_runs_end_to_end(tmp_path):
    """tools/demo_synthetic.py: the call sequence of demo/build_submaps.py:125-141 (dry-run System -> per-submap
    Mapper.mapping -> save_mesh -> torch.save(grid_atlas)) then demo/align_submaps.py:240-317 (torch.load -> perturb
    submap poses -> Fuser.align -> metrics) on synthetic RGB-D frames, through the ``grid_opt`` import names."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "demo_synthetic.py"), "--save_dir", str(tmp_path),
                          "--quick"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert os.path.exists(tmp_path / "grid_atlas.pth") and os.path.exists(tmp_path / "alignment_result.json")
    import json
    import math
    res = json.load(open(tmp_path / "alignment_result.json"))
    # (whether the poses improve depends on the maps: with the seeded RANDOM decoder that stands in for the reference's
    # pretrained decoder_indoor.pt the latent spaces of separately trained submaps need not agree -- the alignment loss
    # falls, the pose error need not; what is checked is that the whole sequence runs and reports finite metrics)
    for when in ("before_alignment", "after_alignment", "shared_field_before", "shared_field_after"):
        assert set(res[when]) == {"rmse_tran (cm)", "rmse_deg"} and all(math.isfinite(v) for v in res[when].values())
    # ... and on features two submaps DO agree on (one analytic field of the world sampled into the same atlas file,
    # tools/shared_field.py) the same perturbation, through the same calls, is undone: the demo asserts a 5x drop of
    # both trajectory errors itself; here the numbers
    bef, aft = res["shared_field_before"], res["shared_field_after"]
    assert aft["rmse_tran (cm)"] <= 0.2 * bef["rmse_tran (cm)"] and aft["rmse_deg"] <= 0.2 * bef["rmse_deg"], (bef, aft)
    assert "3 submaps, 12 keyframes" in out.stdout
