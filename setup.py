"""Shim for setuptools < 61 (this image ships 59.6, which does not read the [project] table): the metadata lives in pyproject.toml
only and is handed to setup() from there; newer setuptools read pyproject.toml themselves and get a bare setup()."""
import os

import setuptools

try:
    import tomllib as _toml
except ModuleNotFoundError:      # Python 3.10
    import tomli as _toml

HERE = os.path.dirname(os.path.abspath(__file__))


def metadata_from_pyproject():
    with open(os.path.join(HERE, "pyproject.toml"), "rb") as f:
        doc = _toml.load(f)
    proj, tool = doc["project"], doc["tool"]["setuptools"]
    return dict(
        name=proj["name"], version=proj["version"], description=proj["description"], python_requires=proj["requires-python"],
        install_requires=proj.get("dependencies", []), extras_require=proj.get("optional-dependencies", {}),
        packages=tool["packages"], include_package_data=tool.get("include-package-data", False),
        package_data=tool.get("package-data", {}), data_files=[(k, v) for k, v in tool.get("data-files", {}).items()],
        entry_points={group: [f"{name} = {target}" for name, target in eps.items()] for group, eps in proj.get("entry-points", {}).items()})


if int(setuptools.__version__.split(".")[0]) < 61:
    setuptools.setup(**metadata_from_pyproject())
else:
    setuptools.setup()
