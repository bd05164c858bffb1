"""CPU oracle for the deephisto hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A plain NumPy / torch-CPU restatement of the reference algorithms on the
path named by BASELINE.json `north_star`.
Every function cites the reference file:line it follows (paths relative to
the upstream tree, xubiker/deephisto @ 2025-02-22).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package, and only as the *checker* (or as the
timed CPU baseline) -- never as part of the shipped data path.  The product
(`deephisto_amd/`) must not import `oracle`; tests/test_no_oracle_in_product.py
enforces that.

Parity pinning status
---------------------
* a1-a5, a8 (tile grid, patch views, /255 features, coords, progress,
  logit accumulation + argmax): pinned against the reference's own
  `FullImageDenseSampler` / `ImagePredictorPatched` executed in the build
  container (`oracle/make_golden.py`, fixtures in `tests/golden/`).  The
  reference has no tests or golden vectors of its own.
* a6/a7 (ResNet-18 forward / train step): **parity unpinned** against the
  reference -- the arithmetic lives in `torchvision.models.resnet18`
  (unpinned in environment.yaml, not installed here) and `torch`; the oracle
  is a restatement of the published torchvision ResNet-18 (BasicBlock
  [2,2,2,2]) with `torch.nn.functional` primitives on CPU fp32, anchored on
  the reference call sites models/patch_cls_simple/model.py:5-11,
  train.py:114-172, examples/predict_full_patched.py:66-78.
"""
