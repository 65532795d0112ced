"""Packaging metadata of the outer boundary: the `nerfstudio.method_configs` entry point the reference registers at
R:pyproject.toml:15-16 (`lsenerf = 'lse_nerf.lse_config:lsenerf_method'`) exists for this package too, points at an attribute
path that resolves, and survives a real metadata build (setuptools, offline, into a scratch directory -- nothing is installed)."""
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pyproject():
    import tomli
    with open(os.path.join(ROOT, "pyproject.toml"), "rb") as f:
        return tomli.load(f)


def test_entry_point_is_declared_like_the_references():
    doc = _pyproject()
    eps = doc["project"]["entry-points"]["nerfstudio.method_configs"]
    assert eps == {"lsenerf-amd": "lsenerf_amd.ns_plugin:lsenerf_method"}
    data = doc["tool"]["setuptools"]["package-data"]["lsenerf_amd"]
    assert "liblse_hip.so" in data
    assert doc["tool"]["setuptools"]["data-files"]["include"] == ["include/lse_hip.h"]
    assert os.path.exists(os.path.join(ROOT, "include", "lse_hip.h"))
    # the hot path must not depend on nerfstudio: it is an extra, the entry point is what needs it
    assert not any("nerfstudio" in d for d in doc["project"]["dependencies"])
    assert any(d.startswith("nerfstudio==0.3.2") for d in doc["project"]["optional-dependencies"]["nerfstudio"])


def test_entry_point_target_resolves_up_to_the_missing_nerfstudio():
    """`module:attr` must name an importable module with that (lazy) attribute.  Resolving the attribute builds the
    MethodSpecification, which needs nerfstudio: absent here, so the clear message is what can be checked."""
    target = _pyproject()["project"]["entry-points"]["nerfstudio.method_configs"]["lsenerf-amd"]
    mod_name, attr = target.split(":")
    mod = importlib.import_module(mod_name)
    assert hasattr(mod, "build_method_specification") and hasattr(mod, "__getattr__")
    try:
        import nerfstudio  # noqa: F401
    except ModuleNotFoundError:
        with pytest.raises(ModuleNotFoundError, match="nerfstudio==0.3.2"):
            getattr(mod, attr)
    else:      # pragma: no cover - not in this image
        assert getattr(mod, attr) is not None
    with pytest.raises(AttributeError):
        getattr(mod, "no_such_method")


def test_setuptools_writes_the_entry_point_into_the_metadata(tmp_path):
    """`setup.py egg_info` into a scratch directory (what `pip install -e . --no-deps --no-build-isolation` runs first), then the
    generated entry_points.txt is read back through importlib.metadata."""
    p = subprocess.run([sys.executable, "setup.py", "-q", "egg_info", "--egg-base", str(tmp_path)], cwd=ROOT, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    from importlib import metadata
    dists = list(metadata.distributions(path=[str(tmp_path)]))
    assert len(dists) == 1
    d = dists[0]
    assert d.metadata["Name"] == "lsenerf-amd" and d.version == _pyproject()["project"]["version"]
    eps = [e for e in d.entry_points if e.group == "nerfstudio.method_configs"]
    assert [(e.name, e.value) for e in eps] == [("lsenerf-amd", "lsenerf_amd.ns_plugin:lsenerf_method")]
    mod = importlib.import_module(eps[0].value.split(":")[0])
    assert mod.__name__ == "lsenerf_amd.ns_plugin"
