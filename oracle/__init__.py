"""CPU oracle for the LSENeRF hot path -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference (ubc-vision/LSENeRF) holds no golden vectors, known-answer
tests or fixtures for this path, and the arithmetic lives in third-party
packages that are absent from /root/reference and not importable here:

  * nerfstudio == 0.3.2   (R:pyproject.toml:6)        HashEncoding torch path, MLP, SH, renderers
  * nerfacc    == 0.5.2   (R:environement.yml:139)    traverse_grids, pack_info, volrend
  * tinycudann == 1.7     (R:environement.yml:220)    HashGrid, FullyFusedMLP/CutlassMLP, SH

This package restates their *published* algorithms in plain torch/numpy/C
and anchors on the reference's own call sites (cited per function as ``R:file:line``).
Until a machine with the real stack is available every "matches the reference" claim is a claim
about this restatement.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  Nothing under ``lsenerf_amd/`` imports it; the product path fails loudly when the
HIP library is missing (see ``lsenerf_amd/_lib.py``).
"""
