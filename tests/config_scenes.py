"""BASELINE.json configs[1..4] at their full sizes: the product's seeded procedural workloads
(nexus_amd.workloads) built as tests.scene_helpers.BuiltScene, so the same bytes go to the oracle and to the device."""
from nexus_amd import workloads
from nexus_amd.workloads import procedural_sky  # noqa: F401  (re-exported for the tests)
from tests import scene_helpers as SH


def config2(width=1920, height=1080, nu=1024, nv=512, path_length=8):
    """configs[1]: the bench workload (bench.py renders the identical scene)."""
    return workloads.config2(width, height, nu, nv, path_length, cls=SH.BuiltScene)


def config4(width=1920, height=1080, path_length=8, n_side=10, nu=250, nv=200):
    return workloads.config4(width, height, path_length, n_side, nu, nv, cls=SH.BuiltScene)


def config5(width=3840, height=2160, path_length=16, field=2200, prop_nu=512, prop_nv=256, n_props=16):
    return workloads.config5(width, height, path_length, field, prop_nu, prop_nv, n_props, cls=SH.BuiltScene)
