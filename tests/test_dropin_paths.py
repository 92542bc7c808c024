"""CPU: the drop-in import seam of INTEGRATION.md section 1 -- with ``semanticlidarunc_amd/`` ahead of a reference-like ``src/`` tree
on sys.path, mirrored modules resolve here, modules that are NOT mirrored still resolve to the other tree (merged package
paths), and names a partially mirrored module lacks are re-exported from the shadowed file.  Hermetic: the "reference" is a
small fake tree written to tmp_path (the real one never travels to the GPU box)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shadowing_merges_packages_and_reexports_missing_names(tmp_path):
    src = tmp_path / "src"
    (src / "utils").mkdir(parents=True)
    (src / "models").mkdir()
    (src / "losses").mkdir()
    (src / "utils" / "vis_other.py").write_text("def only_in_reference():\n    return 'utils.vis_other from the other tree'\n")
    (src / "models" / "probability_helper.py").write_text(textwrap.dedent("""
        from utils.vis_other import only_in_reference          # a reference module importing a non-mirrored sibling
        def build_uncertainty_layers():
            return 'reference helper'
        def get_eps_value():
            return 'shadowed: must NOT win over the mirror'
    """))
    (src / "losses" / "dirichlet_losses.py").write_text("def _valid_mask(t, i):\n    return 'reference _valid_mask'\n")
    (src / "dataset").mkdir()
    (src / "dataset" / "utils.py").write_text("def spherical_projection(pc, *a, **k):\n    return 'reference numpy projection'\ndef rotate_z(p, a):\n    return 'ref rotate'\n")
    (src / "models" / "trainer_like.py").write_text(textwrap.dedent("""
        from models.evaluator import IoUEvaluator, UncertaintyAccuracyAggregator
        from models.probability_helper import to_alpha_concentrations_from_shape_and_scale, build_uncertainty_layers, get_eps_value
        from losses.dirichlet_losses import _valid_mask, DirichletMSELoss
        from utils.vis_other import only_in_reference
        from utils.mc_dropout import mc_forward
    """))
    code = textwrap.dedent("""
        import models.trainer_like as t
        import models.evaluator, models.probability_helper, losses.dirichlet_losses, utils.mc_dropout
        assert 'semanticlidarunc_amd' in models.evaluator.__file__ and 'semanticlidarunc_amd' in utils.mc_dropout.__file__
        assert t.build_uncertainty_layers() == 'reference helper' and t._valid_mask(0, 0) == 'reference _valid_mask'
        assert t.get_eps_value() == 1e-8                      # the mirror's own definition wins
        assert t.only_in_reference().startswith('utils.vis_other')
        assert t.DirichletMSELoss.__module__ == 'losses.dirichlet_losses'
        # the projection mirror: re-exports the other helpers, and inside a DataLoader worker hands the call to the shadowed module
        import dataset.utils as du, torch.utils.data as tud
        assert 'semanticlidarunc_amd' in du.__file__ and du.rotate_z(0, 0) == 'ref rotate'
        tud.get_worker_info = lambda: object()
        import warnings
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            assert du.spherical_projection(None) == 'reference numpy projection'
            assert du.spherical_projection(None) == 'reference numpy projection'
        assert len(w) == 1 and 'DataLoader worker' in str(w[0].message) and 'gpu_loader' in str(w[0].message)      # said once, with the way out
        print('ok')
    """)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "semanticlidarunc_amd"), str(src)]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_package_names_are_unaffected():
    import semanticlidarunc_amd.models.probability_helper as ph
    import semanticlidarunc_amd.utils.mc_dropout as mc
    assert ph.get_eps_value() == 1e-8 and callable(mc.mc_forward)
    assert not any(k.startswith("_slu_shadowed_.") for k in sys.modules)
