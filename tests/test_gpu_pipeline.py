"""End-to-end: the reference's example.py flow on synthetic data (examples/pipeline_synthetic.py),
every stage through libalproj_hip.so; the optimisers must recover the hidden camera."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_example_pipeline_recovers_the_camera():
    from alproj_amd import _lib
    _lib.init(0)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "pipeline_synthetic.py")
    spec = importlib.util.spec_from_file_location("pipeline_synthetic", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.run(n=512, w=702, h=468, generations=120, verbose=False)
    assert min(out["gcps"]) > 500
    assert out["reproj_px_initial"] > 20
    assert out["reproj_px_final"] < 1.5, out
    assert out["errors"][2] < out["errors"][0] < 10.0          # Huber losses incl. the 3 % planted outliers
    assert out["raster_filled"] > 0.2 and out["georectified_rows"] > 50_000
