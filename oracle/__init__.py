"""CPU oracle for the spsbrdf-nerf ray-rendering hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (PyTorch-CPU,
fp32 or fp64) of the reference algorithm for the hot path named by
BASELINE.json:north_star.  It is imported only by `tests/`, by
`__graft_entry__.smoke()` and by the `cpu_baseline` leg of `bench.py`, and
there only as the checker / the timed CPU baseline.  The product path
(`brdf_nerf_amd`) never imports it and fails loudly when the HIP library is
missing.

Parity status: PINNED.  The reference ships no tests or golden vectors
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, produced in the build container by `tests/golden/make_goldens.py`
(which imports the reference read-only from /root/reference) and committed as
small `.npz` fixtures under `tests/golden/`.  `tests/test_oracle_golden.py`
checks every oracle function against those fixtures.

Reference files restated (path:line under /root/reference):
  rendering.py:13-91,116-166,225-291          -> oracle/render.py
  models/spsbrdfnerf.py:50-69,71-416,636-757  -> oracle/field.py, oracle/render.py
  models/nerf.py:9-70                         -> oracle/field.py
  BRDF/basic_func.py, RPV.py, Hapke.py, microfacet.py -> oracle/brdf.py
  train_utils.py:28-39,61-78                  -> oracle/field.py, oracle/render.py
  metrics.py:39-61,82-161                     -> oracle/losses.py
"""
from .config import FieldConfig  # noqa: F401
